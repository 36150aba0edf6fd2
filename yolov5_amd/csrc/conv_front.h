// The first three layers of the YOLOv5 backbone as ONE persistent launch:
//     0.Conv  k6 s2 p2   3 -> C0   (models/yolov5s.yaml:17, models/common.py:74-92)          read straight from the NCHW fp16 batch
//     1.Conv  k3 s2 p1  C0 -> C1   (yolov5s.yaml:18)
//     2.C3.cv1 + 2.C3.cv2  1x1  C1 -> 2 c_   (common.py:246: both read 1.Conv's output and nothing else does; one GEMM with N = 2 c_)
// each with its folded-BatchNorm bias and SiLU.  The 320^2 x C0 output of the stem (419 MB written + 419 MB read back at yolov5s bs = 64) and the
// 160^2 x C1 output of 1.Conv (210 + 210 MB) never exist in HBM: 157 MB of image go in, 2 x 105 MB of C3 inputs come out.
//
// A workgroup of eight waves owns a TH x TW tile of 1.Conv's output (16 x 16) of one image, three phases per tile:
//   (1) STEM: the (2 TH + 1) x (2 TW + 1) stem pixels under the tile are computed from the (4 TH + 6)-row input patch (staged by LDS-DMA as 3 x 70
//       runs of 80 halfs, out-of-image pieces = zero fill = the stem's padding) -- 35 blocks of 32 pixels x 9 MFMA steps, dealt round-robin to the
//       waves -- bias + SiLU, ZEROED where the stem pixel lies outside the stem image (1.Conv pads ITS input with zeros, not with SiLU(bias)), and
//       written as fp16 into the LDS-resident stem patch in the layout phase (2) reads: 64-byte pixel rows, even and odd columns de-interleaved
//       (a stride-2 tap walk becomes a unit-stride one), row pitch 36 pixels, 16-byte slots XOR-swizzled by (q >> 2) & 3 -- conflict-free
//       ds_read_b128 for every tap of the 3x3 (enumerated over all lanes / taps / waves, scripts note in DESIGN.md);
//   (2) 3x3 s2: every wave multiplies its 4 x 8 sub-tile: nine taps = nine LDS row offsets of the same patch, filter resident in LDS;
//   (3) 1x1: the 3x3's bias + SiLU result goes from accumulator registers to the next MFMA's activation operand WITHOUT touching LDS --
//       v_permlane32_swap exchanges the two 4-channel groups a lane pair holds (MFMA D layout: lane (pixel, g) owns channels 8q + 4g .. + 3;
//       B operand: lane (pixel, g) needs channels 16 ks + 8 g .. + 7) -- and the second epilogue leaves the same way as 16-byte stores.
// Two workgroup barriers per tile (input landed / patch complete); the next tile's input patch flies while phases (2)-(3) run.
#pragma once
#include "conv_igemm.h"

#ifdef Y5_FRONT_TIMING
__device__ unsigned long long y5_front_dbg[64];  // workgroup 0, per wave: input wait + barrier, stem, patch barrier, input issue, 3x3, 1x1 + stores (s_memtime), tiles
#define Y5_FT(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define Y5_FT(var)
#endif

struct Y5FrontParams {
  const void* x;                      // (B, 3, H, W) fp16
  const void* w0; const float* b0;    // stem filter [32][144], k = (c * 6 + kh) * 8 + kw (kw 6, 7 zero); bias [32]
  const void* w1; const float* b1;    // 3x3 filter packed [NPAD1][Kpad1], k = (kh, kw, c), c < 32; bias [NPAD1]
  const void* w2; const float* b2;    // 1x1 filter packed [NPAD2][Kpad2], k = the 3x3's output channel; bias [NPAD2]
  void* y; void* y2;                  // 1x1 output channels [0, split) -> y (pixel stride ldy), [split, C3) -> y2 (pixel stride ld2)
  unsigned x_bytes, w1_bytes, w2_bytes;
  int B, H, W;                        // input image
  int OH0, OW0;                       // stem output (H / 2, W / 2)
  int OH1, OW1;                       // 3x3 output (H / 4, W / 4)
  int Kpad1, Kpad2, C3, split, ldy, ld2;
  int act1, act2;
  int tiles_h, tiles_w;
};

template <int TH, int TW> struct Y5FrontGeom {
  static constexpr int SR = 2 * TH + 1, SC = 2 * TW + 1;      // stem patch rows / columns
  static constexpr int HALF = (SC + 1) / 2;                    // even columns first, then the odd ones
  static constexpr int RS = 36;                                // patch row pitch in pixels (bank-conflict free with the (q >> 2) & 3 swizzle for SC = 33)
  static constexpr int IR = 2 * SR + 4;                        // input rows per channel
  static constexpr int NP = 10;                                // 16-byte pieces (8 columns) per input row: columns [4 ow0 - 8, 4 ow0 + 72)
  static constexpr int NPIECE = 3 * IR * NP;
  static constexpr int NI = (NPIECE + 63) / 64;                // LDS-DMA instructions per input patch
  static constexpr int NIW = (NI + 7) / 8;                     // ... per wave (uniform: surplus instructions load zeros into a dummy slot)
  static constexpr int IN_BYTES = NI * 1024;
  static constexpr int NPX = SR * SC;                          // stem pixels per tile
  static constexpr int NB0 = (NPX + 31) / 32;                  // 32-pixel blocks
  static constexpr int NBW = (NB0 + 7) / 8;                    // ... per wave
  static constexpr int SP_BYTES = SR * RS * 64;
  static_assert(SC == 33, "row pitch / swizzle were chosen for a 33-column patch (TW = 16)");
};

template <int TH, int TW, int NT1, int NT2>
constexpr size_t y5_conv_front_lds_bytes() {
  using G = Y5FrontGeom<TH, TW>;
  return (size_t)G::IN_BYTES + 1024 + G::SP_BYTES + (size_t)NT1 * 32 * 576 + (size_t)NT2 * 32 * (NT1 * 64) + (size_t)(32 + NT1 * 32 + NT2 * 32) * 4;
}

// experiment builds (scripts/front_ablate.sh): -DY5_FR_NOSILU identity instead of SiLU, -DY5_FR_NOSTEMMFMA / -DY5_FR_NOSTEM / -DY5_FR_NOL1 skip parts
#ifdef Y5_FR_NOSILU
#define Y5_FR_SILU(v) (v)
#else
#define Y5_FR_SILU(v) y5_silu(v)
#endif

// D-layout registers of one 32-channel block (16 fp32 per lane: channels 8 q + 4 g + e) -> the two B-operand fragments (k-steps 0 / 1 of the block's 32
// channels: lane (pixel, g) holds channels 16 ks + 8 g .. + 7) after bias-free activation; see the header comment
template <bool ACT>
__device__ __forceinline__ void y5_d_to_b_frags(const float16_t& acc, half8_t (&out)[2]) {
  uint32_t h[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ACT ? Y5_FR_SILU(acc[q * 4 + e]) : acc[q * 4 + e];
    h[q][0] = y5_pack_h2(v[0], v[1]);
    h[q][1] = y5_pack_h2(v[2], v[3]);
  }
  uint4_t f[2];
  y5_swap_to_pixel_vectors(h, f);
  out[0] = __builtin_bit_cast(half8_t, f[0]);
  out[1] = __builtin_bit_cast(half8_t, f[1]);
}

template <int TH, int TW, int NT1, int NT2>
__global__ __launch_bounds__(512, 2)
void y5_conv_front_kernel(const Y5FrontParams p) {
  typedef half_t T;
  using G = Y5FrontGeom<TH, TW>;
  static_assert((TH / 4) * (TW / 8) == 8, "eight waves, one 4 x 8 sub-tile each");
  constexpr int NPAD1 = 32 * NT1, NPAD2 = 32 * NT2;
  constexpr int K2 = 576;                    // bytes per 3x3 filter row in LDS: 9 taps x 32 channels x 2
  constexpr int K2B = NPAD1 * 2;             // bytes per 1x1 filter row in LDS
  constexpr int WX = TW / 8;
  constexpr int RS = G::RS, HALF = G::HALF, IR = G::IR;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* in_lds = smem;                                   // input patch: 3 x IR runs of 160 bytes, + slack of the last instruction
  char* dummy = smem + G::IN_BYTES;                      // 1 KiB: receives the zero fill of surplus LDS-DMA instructions
  char* sp_lds = dummy + 1024;                           // stem patch
  char* w1lds = sp_lds + G::SP_BYTES;
  char* w2lds = w1lds + NPAD1 * K2;
  float* b0lds = reinterpret_cast<float*>(w2lds + NPAD2 * K2B);
  float* b1lds = b0lds + 32;
  float* b2lds = b1lds + NPAD1;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, pl = lane & 31;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  T* __restrict__ yg = static_cast<T*>(p.y);
  T* __restrict__ y2g = static_cast<T*>(p.y2);

  // ---- tile schedule -----------------------------------------------------------------------------------------------------------
  const int Gd = gridDim.x, bid = blockIdx.x;
  const int ntiles = p.B * p.tiles_h * p.tiles_w;
  const int nmine = (ntiles - bid + Gd - 1) / Gd;
  auto tile_origin = [&](int j, int& b, int& oh0, int& ow0) {
    const int t = y5_xcd_remap(bid + j * Gd, ntiles);
    const int tx = t % p.tiles_w, r = t / p.tiles_w;
    const int ty = r % p.tiles_h;
    b = r / p.tiles_h; oh0 = ty * TH; ow0 = tx * TW;
  };

  // ---- per-lane constants ------------------------------------------------------------------------------------------------------
  // (a) input pieces of this wave: instruction I = wave + 8 i covers pieces I * 64 + lane; piece -> (channel, row, 8-column group)
  int pi_rel[G::NIW], pi_rj[G::NIW];
#pragma unroll
  for (int i = 0; i < G::NIW; ++i) {
    const int idx = (wave + 8 * i) * 64 + lane;
    const int run = idx / G::NP, j = idx - run * G::NP;
    const int c = run / IR, row = run - c * IR;
    pi_rel[i] = ((c * p.H + row) * p.W + 8 * j) * 2;
    pi_rj[i] = row | (j << 8) | ((idx < G::NPIECE ? 1 : 0) << 16);
  }
  auto input_offsets = [&](int jt, unsigned (&voff)[G::NIW]) {
    int b, oh0, ow0;
    tile_origin(jt, b, oh0, ow0);
    const int ih0 = 4 * oh0 - 4, col0 = 4 * ow0 - 8;
    const int base = ((b * 3 * p.H + ih0) * p.W + col0) * 2;   // may be negative: only used when the piece is inside the image
#pragma unroll
    for (int i = 0; i < G::NIW; ++i) {
      const int ih = ih0 + (pi_rj[i] & 0xff), col = col0 + 8 * ((pi_rj[i] >> 8) & 0xff);
      const bool ok = (pi_rj[i] >> 16) != 0 && (unsigned)ih < (unsigned)p.H && (unsigned)col < (unsigned)p.W;
      voff[i] = ok ? (unsigned)(base + pi_rel[i]) : Y5_OOB;
    }
  };
  auto issue_piece = [&](int i, unsigned voff) {
    const int I = wave + 8 * i;
    y5_bglds16(xrs, voff, I < G::NI ? in_lds + I * 1024 : dummy);
  };
  auto issue_input = [&](int jt) {
    unsigned voff[G::NIW];
    input_offsets(jt, voff);
#pragma unroll
    for (int i = 0; i < G::NIW; ++i) issue_piece(i, voff[i]);
  };
  // (b) stem blocks of this wave: block = wave + 8 s, pixel m = block * 32 + pl -> (sr, sc)
  int sb_in[G::NBW], sb_out[G::NBW], sb_fl[G::NBW];
#pragma unroll
  for (int s = 0; s < G::NBW; ++s) {
    const int m = (wave + 8 * s) * 32 + pl;
    const int mm = m < G::NPX ? m : 0;
    const int sr = mm / G::SC, sc = mm - sr * G::SC;
    sb_in[s] = (2 * sr) * 160 + 4 * sc + 8 + g * 160;
    const int q = sr * RS + (sc >> 1) + (sc & 1) * HALF;
    sb_out[s] = q * 64 + g * 8;
    sb_fl[s] = ((q >> 2) & 3) | ((sr == 0 ? 1 : 0) << 4) | ((sc == 0 ? 1 : 0) << 5) | ((m < G::NPX ? 1 : 0) << 6);
  }
  // (c) 3x3 fragment reads: lane (pixel pl of the wave's 4 x 8 sub-tile, k-half g), tap t, k-step ks
  const int wy = wave / WX, wx = wave - wy * WX;
  const int oy = wy * 4 + (pl >> 3), ox = wx * 8 + (pl & 7);
  int rdA[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int r = 2 * oy + t / 3, c = 2 * ox + t % 3;
    const int q = r * RS + (c >> 1) + (c & 1) * HALF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rdA[t][ks] = q * 64 + (((ks * 2 + g) ^ ((q >> 2) & 3)) * 16);
  }
  int wsl[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wsl[ks] = pl * K2 + (((ks * 2 + g) ^ ((pl >> 2) & 3)) * 16);
  const int sw2 = NPAD1 >= 64 ? ((pl >> 1) & 7) : ((pl >> 2) & 3);

  // ---- prologue: first input patch, filters + biases into LDS, stem filter fragments into registers -------------------------------
  if (nmine > 0) issue_input(0);
  {
    const y5_rsrc_t w1rs = y5_make_rsrc(p.w1, p.w1_bytes);
    constexpr int WSL = 36;  // 16-byte slots per 3x3 filter row
    constexpr int W1I = NPAD1 * WSL / 64;
    for (int I = wave; I < W1I; I += 8) {
      const int pidx = I * 64 + lane;
      const int n = pidx / WSL, ps = pidx - n * WSL;
      const int src_slot = (ps & ~3) | ((ps & 3) ^ ((n >> 2) & 3));
      y5_bglds16(w1rs, (unsigned)((n * p.Kpad1) * 2 + src_slot * 16), w1lds + I * 1024);
    }
    const y5_rsrc_t w2rs = y5_make_rsrc(p.w2, p.w2_bytes);
    constexpr int NSL2 = NPAD1 / 8, W2I = NPAD2 * NSL2 / 64;
    for (int I = wave; I < W2I; I += 8) {
      const int pidx = I * 64 + lane;
      const int n = pidx / NSL2, ps = pidx - n * NSL2;
      const int sw = NSL2 >= 8 ? ((n >> 1) & 7) : ((n >> 2) & 3);
      y5_bglds16(w2rs, (unsigned)(n * p.Kpad2 * 2 + ((ps ^ sw) * 16)), w2lds + I * 1024);
    }
    for (int i = tid; i < 32; i += 512) b0lds[i] = p.b0[i];
    for (int i = tid; i < NPAD1; i += 512) b1lds[i] = p.b1[i];
    for (int i = tid; i < NPAD2; i += 512) b2lds[i] = p.b2[i];
  }
  half8_t wf0[9];
  {
    const T* wg = static_cast<const T*>(p.w0);
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) wf0[ks] = *reinterpret_cast<const half8_t*>(wg + (size_t)pl * 144 + (2 * ks + g) * 8);
  }
  y5_wait_vm<0>();
  __syncthreads();

  // stores one wave issues per tile (the counted wait at the top of the next tile skips exactly these)
  const bool full_c3 = p.C3 == NPAD2;

#ifdef Y5_FRONT_TIMING
  unsigned long long d_wait = 0, d_stem = 0, d_bar = 0, d_issue = 0, d_l1 = 0, d_pw = 0;
#endif
  for (int ti = 0; ti < nmine; ++ti) {
    int tb, oh0, ow0;
    tile_origin(ti, tb, oh0, ow0);
    Y5_FT(ft0);
    if (ti > 0) {
      // this wave's share of the input patch has landed; the previous tile's stores (issued after it) may still be in flight
      if (full_c3) y5_wait_vm<2 * NT2>();
      else y5_wait_vm<0>();
      __builtin_amdgcn_s_barrier();  // everybody's share; every wave is done reading the previous tile's stem patch
    }
    Y5_FT(ft1);
    // ================================ phase 1: stem ================================================================================
    // Software-pipelined over the wave's blocks: the nine MFMAs of block s+1 (one dependent chain: an issue slot every 32 cycles) are interleaved
    // with the ~100 VALU instructions of block s's epilogue (bias is the chain's initial value; SiLU, fp16 packing, the padding mask), and block
    // s+1's fragment reads are issued ahead of both -- in the first version every block was read -> wait -> 9 MFMAs -> epilogue, the matrix core idle
    // through each epilogue and the vector ALU idle through each chain: 9 000 of a tile's 19 800 cycles (in-kernel phase timing, scripts/front_bench.py).
    auto stem_frags = [&](int s, half8_t (&afr)[9]) {
#pragma unroll
      for (int ks = 0; ks < 9; ++ks) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(in_lds + sb_in[s] + ((ks / 3) * IR + 2 * (ks % 3)) * 160);
        uint4_t raw;
        raw[0] = src[0]; raw[1] = src[1]; raw[2] = src[2]; raw[3] = src[3];
        afr[ks] = __builtin_bit_cast(half8_t, raw);
      }
    };
    auto stem_bias = [&](float16_t& acc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4_t bv = *reinterpret_cast<const float4_t*>(b0lds + q * 8 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q * 4 + e] = bv[e];
      }
    };
    auto stem_mfma = [&](const half8_t (&afr)[9], float16_t& acc) {
#ifdef Y5_FR_NOSTEMMFMA
      acc[0] += (float)afr[0][0] + (float)afr[8][7];
      return;
#endif
#ifdef Y5_FR_ACC2
      // two accumulation chains (k-steps 0-4 on top of the bias, 5-8 from zero), summed at the end
      float16_t acc_b;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_b[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0[ks], afr[ks], acc, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0[5 + ks], afr[5 + ks], acc_b, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0[4], afr[4], acc, 0, 0, 0);
      acc += acc_b;
      return;
#endif
#pragma unroll
      for (int ks = 0; ks < 9; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0[ks], afr[ks], acc, 0, 0, 0);
    };
    auto stem_epilogue = [&](int s, const float16_t& acc) {
      const int fl = sb_fl[s];
      // outside the stem image (row -1 at the top edge, column -1 at the left edge): 1.Conv's zero padding.  Branch-free: the activation is
      // evaluated for every lane (written as `zero ? 0 : silu(v)` hipcc wrapped EACH element in its own exec-mask branch) and the packed result is
      // ANDed with a lane mask; lanes past the patch's last pixel write to the dummy slot instead of being predicated off.
      const uint32_t keep = (((fl & 16) && oh0 == 0) || ((fl & 32) && ow0 == 0)) ? 0u : 0xffffffffu;
      const int fq = fl & 3;
      char* const dst = (fl & 64) ? sp_lds + sb_out[s] : dummy + lane * 8;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = Y5_FR_SILU(acc[r]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2_t o;
        o[0] = y5_pack_h2(v[q * 4 + 0], v[q * 4 + 1]) & keep;
        o[1] = y5_pack_h2(v[q * 4 + 2], v[q * 4 + 3]) & keep;
        *reinterpret_cast<uint2_t*>(dst + ((fl & 64) ? ((q ^ fq) * 16) : 0)) = o;
      }
    };
    // interleave request to the scheduler for one pipelined step: the next block's LDS reads, a first slice of VALU while they are in flight, then
    // one MFMA per ~10 VALU
    auto stem_interleave = [&]() {
#ifndef Y5_EMU
      __builtin_amdgcn_sched_group_barrier(0x100, 22, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
      }
#endif
    };
#ifndef Y5_FR_NOSTEM
    {
      static_assert(G::NBW == 5 && G::NB0 > 32, "pipeline written for four full rounds of blocks + a partial fifth");
      float16_t acc_c, acc_n;
      half8_t afr[9];
      stem_frags(0, afr);
      stem_bias(acc_c);
      stem_mfma(afr, acc_c);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        stem_frags(s + 1, afr);
        stem_bias(acc_n);
        stem_mfma(afr, acc_n);
        stem_epilogue(s, acc_c);
        stem_interleave();
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        acc_c = acc_n;
      }
      if (wave + 32 < G::NB0) {  // wave-uniform: the partial fifth round
        stem_frags(4, afr);
        stem_bias(acc_n);
        stem_mfma(afr, acc_n);
        stem_epilogue(3, acc_c);
        stem_interleave();
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        stem_epilogue(4, acc_n);
      } else {
        stem_epilogue(3, acc_c);
      }
    }
#endif
    Y5_FT(ft2);
    // the patch is complete and nobody reads the input buffer any more
#ifndef Y5_EMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    Y5_FT(ft3);
    // the next tile's input patch: addresses now, the LDS-DMA instructions between the taps of the 3x3 (an LDS-DMA issue holds its wave for 100-400
    // cycles in a busy phase -- 1 200-1 900 cycles per tile when the five were issued back to back here -- time the matrix core now spends on MFMAs)
    unsigned in_voff[G::NIW];
    const bool more_tiles = ti + 1 < nmine;
    if (more_tiles) input_offsets(ti + 1, in_voff);
    Y5_FT(ft4);
#ifdef Y5_FR_NOL1
    if (more_tiles)
      for (int i = 0; i < G::NIW; ++i) issue_piece(i, in_voff[i]);
    y5_wait_vm<0>();
    continue;
#endif
    // ================================ phase 2: 3x3 stride 2 ========================================================================
    float16_t acc1[NT1];
#pragma unroll
    for (int j = 0; j < NT1; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4_t bv = *reinterpret_cast<const float4_t*>(b1lds + j * 32 + q * 8 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1[j][q * 4 + e] = bv[e];
      }
    static_assert(G::NIW <= 5, "one LDS-DMA instruction behind each of the taps 1, 3, 5, 7, 8");
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const half8_t af = *reinterpret_cast<const half8_t*>(sp_lds + rdA[t][ks]);
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
          const half8_t wf = *reinterpret_cast<const half8_t*>(w1lds + j * 32 * K2 + t * 64 + wsl[ks]);
          acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc1[j], 0, 0, 0);
        }
      }
      constexpr int slot_of_tap[9] = {-1, 0, -1, 1, -1, 2, -1, 3, 4};
      if (slot_of_tap[t] >= 0 && slot_of_tap[t] < G::NIW && more_tiles) {
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        issue_piece(slot_of_tap[t], in_voff[slot_of_tap[t]]);
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
#ifdef Y5_FRONT_TIMING
    asm volatile("" :: "v"(acc1[0][0]));
#endif
    Y5_FT(ft5);
    // ================================ phase 3: 1x1 on the register-resident result ==================================================
    half8_t bf[NT1][2];
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
      if (p.act1) y5_d_to_b_frags<true>(acc1[j], bf[j]);
      else y5_d_to_b_frags<false>(acc1[j], bf[j]);
    }
    float16_t acc2[NT2];
#pragma unroll
    for (int j = 0; j < NT2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4_t bv = *reinterpret_cast<const float4_t*>(b2lds + j * 32 + q * 8 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2[j][q * 4 + e] = bv[e];
      }
#pragma unroll
    for (int kk = 0; kk < 2 * NT1; ++kk)
#pragma unroll
      for (int j = 0; j < NT2; ++j) {
        const half8_t wf = *reinterpret_cast<const half8_t*>(w2lds + (j * 32 + pl) * K2B + (((kk * 2 + g) ^ sw2) * 16));
        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bf[kk >> 1][kk & 1], acc2[j], 0, 0, 0);
      }
    // second epilogue: SiLU -> fp16 -> lane-pair exchange -> 16-byte stores (lane (pixel, g): channels 32 j + 16 ks + 8 g .. + 7)
    const size_t m = ((size_t)tb * p.OH1 + oh0 + oy) * p.OW1 + ow0 + ox;
#pragma unroll
    for (int j = 0; j < NT2; ++j) {
      half8_t of[2];
      if (p.act2) y5_d_to_b_frags<true>(acc2[j], of);
      else y5_d_to_b_frags<false>(acc2[j], of);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int n0 = j * 32 + ks * 16;        // wave-uniform: the whole instruction is issued or not
        if (n0 < p.C3) {
          const int n = n0 + g * 8;
          T* d = n < p.split ? yg + m * p.ldy + n : y2g + m * p.ld2 + (n - p.split);
          *reinterpret_cast<uint4_t*>(d) = __builtin_bit_cast(uint4_t, of[ks]);
          Y5_EMU_VM_OP(true);
        }
      }
    }
#ifdef Y5_FRONT_TIMING
    {
      const unsigned long long ft6 = __builtin_amdgcn_s_memtime();
      d_wait += ft1 - ft0; d_stem += ft2 - ft1; d_bar += ft3 - ft2; d_issue += ft4 - ft3; d_l1 += ft5 - ft4; d_pw += ft6 - ft5;
    }
#endif
  }
#ifdef Y5_FRONT_TIMING
  if (lane == 0 && blockIdx.x == 0) {
    unsigned long long* o = y5_front_dbg + wave * 8;
    o[0] = d_wait; o[1] = d_stem; o[2] = d_bar; o[3] = d_issue; o[4] = d_l1; o[5] = d_pw; o[6] = nmine;
  }
#endif
}
