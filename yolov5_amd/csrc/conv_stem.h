// Stem convolution of the YOLOv5 backbone (models/yolov5s.yaml:17 `Conv [64, 6, 2, 2]`: k6 s2 p2, 3 input channels,
// models/common.py:74-92) read STRAIGHT from the caller's NCHW fp16 batch -- the NCHW->NHWC repack of the input
// contract (train.py:379, detect.py:206-210) never touches HBM -- and written as NHWC with fused bias + SiLU.
//
// GEMM view: D[n][m] = sum_k W[n][k] * A[m][k], m = output pixel, k = (c, kh, kw8): 18 (c,kh) runs x 8 taps (kw 6,7 are
// zero in W) = 144, nine mfma_f32_32x32x16_f16 steps.  A wave tile = 32 consecutive output pixels of one output row;
// its receptive field is 18 runs (3 channels x 6 input rows) of 80 contiguous input columns [2*ow0-8, 2*ow0+72) --
// ten 16-byte pieces per run, each piece entirely inside or entirely outside the image because W % 8 == 0 -- staged
// by LDS-DMA (out-of-image pieces come from the zero page).  Lane p's B-fragment for run r is the 8 halfs at columns
// 2p+6 .. 2p+13 of that run: four consecutive dwords, bank-conflict free.  The filter fragments (9 x NT x 16 B per
// lane) stay in registers for the lifetime of the persistent workgroup.  Streaming structure as conv_pw.h:
// wave-private S-stage rings, counted vmcnt, no barriers in the loop, epilogue transposed through the vacated stage.
#pragma once
#include "conv_pw.h"

struct Y5StemParams {
  const void* x;      // (B, 3, H, W) fp16
  const void* w;      // [Npad][144] fp16, k = (c*6 + kh)*8 + kw
  const float* bias;  // [Npad]
  void* y;            // NHWC, pixel stride ldy
  const void* zero;
  int B, H, W, OH, OW, C2, ldy;
  int tiles_per_row;  // OW / 32
  int nwt;            // B * OH * tiles_per_row wave tiles
};

template <int NT, int S>
constexpr size_t y5_conv_stem_lds_bytes() {
  constexpr int STAGE = 3072 > 32 * NT * 64 ? 3072 : 32 * NT * 64;
  return (size_t)NT * 32 * 4 + (size_t)4 * S * STAGE;
}

// (launch bound 2: with a 256-register budget the compiler keeps the accumulators in VGPRs -- at a 512 budget it parks them in AGPRs and the epilogue,
// which is this kernel's issue bound, pays 16 v_accvgpr_read per tile; LDS still limits a CU to three workgroups)
// RAW: the convolution's plain output (no bias, no activation) -- the train-mode forward, whose BatchNorm + SiLU follow as their own passes
template <int NT, int S, bool RAW = false>
__global__ __launch_bounds__(256, 2)
void y5_conv_stem_kernel(const Y5StemParams p) {
  typedef half_t T;
  constexpr int NPAD = 32 * NT;
  constexpr int STAGE = 3072 > 32 * NPAD * 2 ? 3072 : 32 * NPAD * 2;
  constexpr int LP = 3;
  constexpr int SPR = NPAD / 8, RPP = 64 / SPR, NPASS = 32 / RPP, SP = NPASS;
  constexpr int SWM = SPR >= 8 ? 7 : SPR - 1;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* blds = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = smem + NPAD * 4 + wave * (S * STAGE);

  const char* __restrict__ xg = static_cast<const char*>(p.x);
  const T* __restrict__ wg = static_cast<const T*>(p.w);
  T* __restrict__ yg = static_cast<T*>(p.y);
  const char* zero = static_cast<const char*>(p.zero);

  const int g = lane >> 5, frow = lane & 31;
  // filter fragments: lane (n = frow, g) needs k = ks*16 + g*8 .. +7 = run (2ks+g), 16 contiguous bytes of row n
  half8_t wf[NT][9];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int ks = 0; ks < 9; ++ks)
      wf[j][ks] = *reinterpret_cast<const half8_t*>(wg + (size_t)(j * 32 + frow) * 144 + (2 * ks + g) * 8);
  for (int i = tid; i < NPAD; i += 256) blds[i] = p.bias[i];
  __syncthreads();

  // per-lane description of its three 16-byte pieces (index q*64 + lane of 180): run (c, kh) and piece j inside the run
  int pc_off[3], pc_kh[3], pc_j[3];
  bool pc_ok[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int idx = q * 64 + lane;
    const int run = idx / 10, j = idx - run * 10;
    const int c = run / 6, kh = run - c * 6;
    pc_ok[q] = idx < 180;
    pc_kh[q] = kh;
    pc_j[q] = j;
    pc_off[q] = ((c * p.H + kh) * p.W + 8 * j) * 2;  // bytes relative to (b, c=0, ih = 2oh-2, col = 2ow0-8)
  }

  const int G = gridDim.x, bid = blockIdx.x;
  const int nbt = (p.nwt + 3) >> 2;
  const int nmine = (nbt - bid + G - 1) / G;
  auto tile_id = [&](int j) { return y5_xcd_remap(bid + j * G, nbt) * 4 + wave; };
  int nw = nmine;
  if (nw > 0 && tile_id(nw - 1) >= p.nwt) --nw;

  auto issue = [&](int jt, int buf) {
    const int t = tile_id(jt);
    const int owt = t % p.tiles_per_row;
    const int r = t / p.tiles_per_row;
    const int oh = r % p.OH, b = r / p.OH;
    const int ih0 = 2 * oh - 2, col0 = 64 * owt - 8;
    const long long base = (((long long)b * 3 * p.H + ih0) * p.W + col0) * 2;
    char* dst = ring + buf * STAGE;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int ih = ih0 + pc_kh[q], col = col0 + 8 * pc_j[q];
      const bool ok = pc_ok[q] && (unsigned)ih < (unsigned)p.H && (unsigned)col < (unsigned)p.W;
      y5_glds16(ok ? xg + base + pc_off[q] : zero, dst + q * 1024);
    }
  };

  const int orow = lane / SPR, oslot = lane % SPR;
  for (int s = 0; s < S; ++s)
    if (s < nw) issue(s, s);

  int buf = 0;
  for (int i = 0; i < nw; ++i) {
    if (i + S - 1 >= nw) {
      y5_wait_vm<0>();
    } else if (i < S - 1) {
      switch (i) {
        case 0: y5_wait_vm<(S - 1) * LP>(); break;
        case 1: y5_wait_vm<(S - 1) * LP + SP>(); break;
        case 2: y5_wait_vm<(S - 1) * LP + 2 * SP>(); break;
        default: y5_wait_vm<(S - 1) * LP>(); break;
      }
    } else {
      y5_wait_vm<(S - 1) * (LP + SP)>();
    }
    __builtin_amdgcn_wave_barrier();
    char* st = ring + buf * STAGE;
    // NT == 1: ONE accumulation chain (a dependent MFMA reads its SrcC from the previous one's result path without a bubble worth the
    // 16 v_accvgpr_read + 8 v_pk_add_f32 a second accumulator adds to an epilogue that is this kernel's issue bound); NT == 2: the two
    // output blocks interleave, a second accumulator per block buys nothing either
    float16_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // all nine activation fragments first (36 dwords in flight), then the nine MFMAs back to back: left to the scheduler the loop was
    // read -> lgkmcnt(0) -> MFMA per k-step, an LDS round trip in front of every multiply
    half8_t afr[9];
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(st + (2 * ks + g) * 160 + 4 * frow + 12);
      uint4_t raw;
      raw[0] = src[0]; raw[1] = src[1]; raw[2] = src[2]; raw[3] = src[3];
      afr[ks] = __builtin_bit_cast(half8_t, raw);
    }
#ifndef Y5_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][ks], afr[ks], acc[j], 0, 0, 0);
    }
#ifndef Y5_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
    // epilogue: bias + SiLU -> scratch (vacated stage) -> full-row 16-byte stores
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4_t bv = *reinterpret_cast<const float4_t*>(blds + j * 32 + q * 8 + g * 4);
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = RAW ? (half_t)acc[j][q * 4 + e] : (half_t)y5_silu(acc[j][q * 4 + e] + bv[e]);
        const int slot = j * 4 + q;
        *reinterpret_cast<half4_t*>(st + frow * (NPAD * 2) + ((slot ^ (frow & SWM)) * 16) + g * 8) = o;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long long m0 = (long long)tile_id(i) * 32;  // tiles enumerate (b, oh, owt) row-major == pixel order
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + orow;
      const uint4_t raw = *reinterpret_cast<const uint4_t*>(st + row * (NPAD * 2) + ((oslot ^ (row & SWM)) * 16));
      const int n = oslot * 8;
      if (n < p.C2) *reinterpret_cast<uint4_t*>(yg + (size_t)(m0 + row) * p.ldy + n) = raw;
      Y5_EMU_VM_OP(true);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (i + S < nw) issue(i + S, buf);
    buf = buf + 1 == S ? 0 : buf + 1;
  }
}
