// Pointwise (1x1 s1) convolution for the DEEP levels with K STREAMED and the whole (<= 256-channel) N tile owned by one workgroup -- the main loop of
// conv_headk.h (the fused Detect head of P4 / P5, which turned out to multiply its 256 x 256-channel layer in about the time the implicit GEMM needs for
// the convolution alone) with a plain epilogue: bias + SiLU -> per-wave LDS transpose -> 16-byte row-contiguous stores (optionally split between two
// destinations: C3's cv1 / cv2 halves).  models/common.py:74-92 `Conv.forward_fuse` for the 1x1 layers of P4 / P5 (C3.cv1+cv2, C3.cv3, 10 / 14.Conv ...).
//   * workgroup tile: 256 pixels x NT * 32 output channels (NT = 8: 256, NT = 4: 128); wave w owns pixels 32 w .. 32 w + 31 and ALL channels of the tile:
//     one activation fragment read serves NT MFMAs; the filter chunk is shared by the eight waves through LDS;
//   * per 32-channel K chunk: x tile 256 x 64 B + filter chunk NT * 32 x 64 B through an NS-stage ring (LDS-DMA, counted vmcnt, one barrier per chunk);
//   * an activation element crosses L2 -> LDS exactly once per N tile (the implicit GEMM's 128 x 128 tiles read it N / 128 times).
#pragma once
#include "conv_igemm.h"

template <int NS_, int NT_>
struct Y5PwkGeom {
  static constexpr int NW = 8, NS = NS_, NT = NT_, BM = NW * 32, BN = NT * 32;
  static constexpr int XS = BM * 64, WS = BN * 64, STAGE = XS + WS;
  static constexpr int XPW = BM / 16 / NW, WPW = (BN / 16 + NW - 1) / NW, PPW = XPW + WPW;
  static constexpr int HALF = NT >= 4 ? NT / 2 : NT;                 // column blocks per epilogue pass
  static constexpr int PITCH = HALF * 64 + 16, PARK = 32 * PITCH;   // per-wave transposition scratch (inside the idle ring)
  static constexpr size_t RING = (size_t)NS * STAGE, OFF_BIAS = RING, OFF_DUMMY = OFF_BIAS + BN * 4, LDS = OFF_DUMMY + 1024;
  static_assert((size_t)NW * PARK <= RING && LDS <= 160 * 1024 && NS >= 3, "LDS budget");
};

template <int NS_, int NT_>
__global__ __launch_bounds__(512, 2)
void y5_conv_pwk_kernel(const Y5ConvParams p) {
  typedef half_t T;
  using Gm = Y5PwkGeom<NS_, NT_>;
  constexpr int NW = Gm::NW, NS = Gm::NS, NT = Gm::NT, STAGE = Gm::STAGE, XS = Gm::XS, XPW = Gm::XPW, WPW = Gm::WPW, PPW = Gm::PPW;
  constexpr int HALF = Gm::HALF, PITCH = Gm::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const blds = reinterpret_cast<float*>(smem + Gm::OFF_BIAS);
  char* const dummy = smem + Gm::OFF_DUMMY;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, frow = lane & 31;
  const int NK = p.C1 >> 5;
  const int m00 = blockIdx.x * Gm::BM, n00 = blockIdx.y * Gm::BN;
  const int m0 = m00 + wave * 32;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t wrs = y5_make_rsrc(p.w, p.w_bytes);
  for (int i = tid; i < Gm::BN; i += NW * 64) blds[i] = n00 + i < p.Npad ? p.bias[n00 + i] : 0.f;

  unsigned x_off[XPW], w_off[WPW];
#pragma unroll
  for (int q = 0; q < XPW; ++q) {
    const int idx = (q * NW + wave) * 64 + lane;
    const int row = idx >> 2, ss = (idx & 3) ^ ((row >> 2) & 3);
    x_off[q] = m00 + row < p.M ? (unsigned)(((long long)(m00 + row) * p.ldx) * 2 + ss * 16) : Y5_OOB;
  }
#pragma unroll
  for (int q = 0; q < WPW; ++q) {
    const int idx = (q * NW + wave) * 64 + lane;
    const int row = idx >> 2, ss = (idx & 3) ^ ((row >> 2) & 3);
    w_off[q] = (q * NW + wave) * 16 < Gm::BN && n00 + row < p.Npad ? (unsigned)(((n00 + row) * p.Kpad) * 2 + ss * 16) : Y5_OOB;
  }
  auto issue = [&](int c) {
    char* st = smem + (c % NS) * STAGE;
#pragma unroll
    for (int q = 0; q < XPW; ++q) y5_bglds16(xrs, x_off[q] == Y5_OOB ? Y5_OOB : x_off[q] + (unsigned)(c * 64), st + (q * NW + wave) * 1024);
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
      if ((q * NW + wave) * 16 < Gm::BN) y5_bglds16(wrs, w_off[q] == Y5_OOB ? Y5_OOB : w_off[q] + (unsigned)(c * 64), st + XS + (q * NW + wave) * 1024);
      else y5_bglds16_dummy(wrs, dummy);
    }
  };
  auto issue_dummy = [&]() {
#pragma unroll
    for (int q = 0; q < PPW; ++q) y5_bglds16_dummy(xrs, dummy);
  };

  const int hp = wave * 32 + frow;
  const int a_rd = (hp << 6) | ((g ^ ((hp >> 2) & 3)) << 4);
  const int w_rd = XS + (frow << 6) + ((g ^ ((frow >> 2) & 3)) << 4);   // + j * 32 rows
  float16_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#pragma unroll
  for (int c = 0; c < NS - 1; ++c)
    if (c < NK) issue(c);
    else issue_dummy();
  for (int c = 0; c < NK; ++c) {
    y5_wait_vm<(NS - 2) * PPW>();   // chunk c has landed (chunks c+1 .. c+NS-2 may be in flight; dummies past the end keep the count constant)
    __builtin_amdgcn_s_barrier();
    if (c + NS - 1 < NK) issue(c + NS - 1);   // into the stage chunk c - 1 occupied: every wave finished reading it before this barrier
    else issue_dummy();
    const char* st = smem + (c % NS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const half8_t af = *reinterpret_cast<const half8_t*>(st + (a_rd ^ (ks * 32)));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const half8_t wf = *reinterpret_cast<const half8_t*>(st + ((w_rd + j * 32 * 64) ^ (ks * 32)));
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc[j], 0, 0, 0);
      }
    }
  }
  y5_wait_vm<0>();
  __syncthreads();   // every wave is done with the ring: a wave's slice of it becomes its transposition scratch
  if (m0 >= p.M) return;

  // ---- epilogue: bias + act -> scratch [pixel][HALF * 32 channels] -> 16-byte row-contiguous stores, one half of the channels at a time ---------------
  char* sc = smem + wave * Gm::PARK;
  T* yg = static_cast<T*>(p.y);
  T* y2g = static_cast<T*>(p.y2);
  constexpr int SPR = HALF * 4;          // 16-byte slots per scratch row
  constexpr int RPP = 64 / SPR;          // rows per store pass
  const int orow = lane / SPR, oslot = lane % SPR;
#pragma unroll
  for (int h = 0; h < NT / HALF; ++h) {
#pragma unroll
    for (int jj = 0; jj < HALF; ++jj) {
      const int j = h * HALF + jj;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4_t bv = *reinterpret_cast<const float4_t*>(blds + j * 32 + q * 8 + g * 4);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = acc[j][q * 4 + e] + bv[e]; v[e] = p.act ? y5_silu(t) : t; }
        uint2_t o;
        o[0] = y5_pack_h2(v[0], v[1]);
        o[1] = y5_pack_h2(v[2], v[3]);
        *reinterpret_cast<uint2_t*>(sc + frow * PITCH + (jj * 32 + q * 8 + g * 4) * 2) = o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ps = 0; ps < 32 / RPP; ++ps) {
      const int row = ps * RPP + orow;
      const int m = m0 + row, n = n00 + h * HALF * 32 + oslot * 8;
      if (m < p.M && n < p.C2) {
        const uint4_t raw = *reinterpret_cast<const uint4_t*>(sc + row * PITCH + oslot * 16);
        if (p.split_n) {
          T* d = n < p.split_n ? yg + (size_t)m * p.ldy + n : y2g + (size_t)m * p.ld2 + (n - p.split_n);
          *reinterpret_cast<uint4_t*>(d) = raw;
        } else {
          *reinterpret_cast<uint4_t*>(yg + (size_t)m * p.ldy + n) = raw;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
