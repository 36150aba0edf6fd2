// 3x3 (pad 1, stride 1 or 2) convolution for the HBM-bound, small-channel layers of the YOLOv5 backbone
// (Conv 1 `3x3 s2 32->64 @320`, Bottleneck.cv2 `3x3 32->32 @160`, `3x3 64->64 @80`: models/yolov5s.yaml:18-21,
// models/common.py:164-181).  Streaming structure of conv_pw.h -- filter resident in LDS, wave-private rings of
// LDS-DMA stages retired with counted vmcnt, no barriers in the loop -- applied to a spatial convolution:
//   * a wave tile is a 4x8 block of output pixels; its receptive field ((3+3s) x (7s+3) input pixels, all C1
//     channels) is staged ONCE per tile as whole pixels (C1*2 bytes each, 16-byte pieces XOR-swizzled on the source
//     side), so every input element crosses L2->LDS 1.9x (s=1) instead of the 9x of an im2col stage;
//   * the nine taps are nine different LDS row offsets of the same stage: the MFMA B-fragment of lane (pixel p, k-half g)
//     for tap (kh,kw) is one ds_read_b128 at a per-lane address computed once per kernel;
//   * padding taps / image borders come from buffer-addressed LDS-DMA with out-of-range offsets (zero fill);
//   * the Bottleneck residual (common.py:181) is fetched with ordinary loads at tile start and added in the epilogue.
#pragma once
#include "conv_pw.h"

#ifdef Y5_K3_TIMING
__device__ unsigned long long y5_k3_dbg[64];  // workgroup 0, per wave: stage wait, residual issue, MFMA loop, epilogue + stores, refill issue (s_memtime), tiles
#endif

template <int C1, int NT, int SH, int S, int NT2 = 0, bool WREG = false, int NWV = 4>
constexpr size_t y5_conv_k3_lds_bytes() {
  constexpr int RH = 3 * SH + 3, RW = 7 * SH + 3, NSL = C1 / 8;
  constexpr int NI = (RH * RW * NSL + 63) / 64;
  return (WREG ? 0 : (size_t)NT * 32 * 9 * C1 * 2) + (size_t)NT * 32 * 4 + (size_t)NWV * S * NI * 1024 + (size_t)NT2 * 32 * (NT * 32 * 2 + 4);
}

// NT2 > 0 (PW2): a pointwise convolution with NT2 * 32 (padded) output channels is applied to every finished tile before it leaves the wave --
// `Conv(3x3, s2)` followed by the merged `C3.cv1 + C3.cv2` GEMM (models/yolo.py walks 1.Conv -> 2.C3; common.py:246 reads the Conv's output
// through cv1 and cv2 only), so the 3x3's output never reaches HBM: its bias + SiLU result is parked in the vacated stage in exactly the
// layout an MFMA activation fragment is read from (pixel rows, 16-byte slots XOR-swizzled by the row), multiplied with the second filter
// (resident in LDS beside the first), and the SECOND epilogue's result is what the tile stores (split over two destinations like the
// split store of conv_igemm.h).
// WREG: the filter's MFMA fragments (9 taps x C1/16 k-steps x NT blocks x 16 bytes per lane) live in REGISTERS for the lifetime of the workgroup
// instead of LDS.  At C1 = 64 the LDS-resident filter (74 KB) leaves room for one workgroup of four waves per CU and every MFMA waits for a filter
// fragment read (72 of the 108 ds_read_b128 per tile: the loop ran at ~30 % matrix-core occupancy, 62 us for a layer whose HBM floor is 31 us); one
// wave per SIMD has 512 registers, 288 of which hold the filter here, and LDS carries only the activation stages (more of them).
// NWV: waves per workgroup (4 or 8).  The in-kernel phase timing (scripts/k3_timing.py) shows a wave spending ~55 % of a tile outside the MFMA loop -- every
// vector-memory instruction (stage refill, residual loads, stores) holds it 100-160 cycles at issue, plus the epilogue -- and with the filter in LDS only one
// workgroup fits a CU: at 4 waves every SIMD idles through all of that.  Eight waves with ONE stage each share the same filter copy: two waves per SIMD,
// one's memory phases under the other's MFMAs ("resident waves beat prefetch depth", as measured for conv_bneck.h).
template <int C1, int NT, int SH, int S, bool RES, bool ACT = true, int NT2 = 0, bool WREG = false, int NWV = 4>
__global__ __launch_bounds__(NWV * 64)
void y5_conv_k3_kernel(const Y5ConvParams p) {
  typedef half_t T;
  constexpr int NPAD = 32 * NT;
  constexpr int TR = 4, TC = 8;                       // wave tile: 4 x 8 output pixels
  constexpr int RH = (TR - 1) * SH + 3, RW = (TC - 1) * SH + 3;  // receptive field in input pixels
  constexpr int NSL = C1 / 8;                         // 16-byte slots per pixel
  constexpr int ROWB = C1 * 2;                        // bytes per staged pixel
  constexpr int NPIECE = RH * RW * NSL;
  constexpr int NI = (NPIECE + 63) / 64;              // LDS-DMA instructions per tile
  constexpr int STAGE = NI * 1024;
  constexpr int K2 = 9 * C1 * 2;                      // bytes per filter row in LDS
  constexpr int WSL = 9 * NSL;                        // 16-byte slots per filter row
  constexpr int W_BYTES = WREG ? 0 : NPAD * K2;
  constexpr int KS = C1 / 16;                         // MFMA k-steps per tap
  constexpr int SPR = NPAD / 8, RPP = 64 / SPR, NPASS = 32 / RPP;
  constexpr int SWM = SPR >= 8 ? 7 : SPR - 1;
  constexpr int NPAD2 = 32 * NT2;                     // PW2: padded output channels of the fused 1x1
  constexpr int SPR2 = NT2 ? NPAD2 / 8 : 8, RPP2 = 64 / SPR2, NPASS2 = 32 / RPP2, SWM2 = SPR2 >= 8 ? 7 : SPR2 - 1;
  constexpr int LP = NI, SP = NT2 ? NPASS2 : NPASS, RP = RES ? NPASS : 0;
  static_assert(STAGE >= 32 * NPAD * 2, "epilogue scratch must fit in a stage");
  static_assert(NT2 == 0 || (!RES && STAGE >= 32 * NPAD2 * 2), "fused 1x1: no residual, second scratch must fit in a stage");
  constexpr int K2B = NPAD * 2;                       // bytes per row of the second filter in LDS (k = the 3x3's padded output channels)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wlds = smem;
  float* blds = reinterpret_cast<float*>(smem + W_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* w2lds = smem + W_BYTES + NPAD * 4;
  float* b2lds = reinterpret_cast<float*>(w2lds + NPAD2 * K2B);
  char* ring = smem + W_BYTES + NPAD * 4 + NPAD2 * (K2B + 4) + wave * (S * STAGE);

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t wrs = y5_make_rsrc(p.w, p.w_bytes);
  T* __restrict__ yg = static_cast<T*>(p.y);
  const T* rg = static_cast<const T*>(p.res);  // may alias y (in-place residual)

  auto fsw = [](int q) { return C1 == 32 ? ((q >> 2) & 3) : ((q >> 1) & 7); };  // pixel-row swizzle (64 / 128-byte rows)

  // ---- prologue: filter (swizzled per row like the activation rows) + bias into LDS ---------------------------
  {
    constexpr int WI = WREG ? 0 : NPAD * WSL / 64;
    for (int I = wave; I < WI; I += NWV) {
      const int pidx = I * 64 + lane;
      const int n = pidx / WSL, ps = pidx - n * WSL;
      const int src_slot = (ps & ~(NSL - 1)) | ((ps & (NSL - 1)) ^ fsw(n));
      y5_bglds16(wrs, (unsigned)((n * p.Kpad) * 2 + src_slot * 16), wlds + I * 1024);
    }
    for (int i = tid; i < NPAD; i += NWV * 64) blds[i] = p.bias[i];
    if constexpr (NT2 > 0) {
      const y5_rsrc_t w2rs = y5_make_rsrc(p.pw2_w, p.pw2_w_bytes);
      constexpr int NSL2 = NPAD / 8, W2I = NPAD2 * NSL2 / 64;
      for (int I = wave; I < W2I; I += NWV) {
        const int pidx = I * 64 + lane;
        const int n = pidx / NSL2, ps = pidx - n * NSL2;
        const int sw = NSL2 >= 8 ? ((n >> 1) & 7) : ((n >> 2) & 3);
        y5_bglds16(w2rs, (unsigned)(n * p.pw2_kpad * 2 + ((ps ^ sw) * 16)), w2lds + I * 1024);
      }
      for (int i = tid; i < NPAD2; i += NWV * 64) b2lds[i] = i < p.pw2_npad ? p.pw2_bias[i] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }

  // ---- per-lane constants -----------------------------------------------------------------------------------
  // (a) the NI pieces this lane stages: pixel q = (r, cq) of the receptive field, channel slot cs
  int pc_rel[NI];   // byte offset relative to the tile's (ih0, iw0) pixel, source-swizzled slot included
  int pc_rc[NI];    // r | cq << 8 | valid << 16
#pragma unroll
  for (int I = 0; I < NI; ++I) {
    const int idx = I * 64 + lane;
    const int q = idx / NSL, cs = idx - q * NSL;
    const int r = q / RW, cq = q - r * RW;
    pc_rel[I] = ((r * p.W + cq) * p.ldx) * 2 + ((cs ^ fsw(q)) * 16);
    pc_rc[I] = r | (cq << 8) | ((idx < NPIECE ? 1 : 0) << 16);
  }
  // (b) fragment read addresses: lane (pixel pl = lane & 31, k-half g), tap t, k-step ks
  const int g = lane >> 5, pl = lane & 31;
  const int q0 = ((pl >> 3) * SH) * RW + (pl & 7) * SH;
  int rd[9][KS];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int q = q0 + (t / 3) * RW + (t % 3);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rd[t][ks] = q * ROWB + (((ks * 2 + g) ^ fsw(q)) * 16);
  }
  int wsl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wsl[ks] = pl * K2 + (((ks * 2 + g) ^ fsw(pl)) * 16);
  half8_t wreg[WREG ? 9 : 1][WREG ? KS : 1][WREG ? NT : 1];
  if constexpr (WREG) {
    const T* wg = static_cast<const T*>(p.w);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          wreg[t][ks][j] = *reinterpret_cast<const half8_t*>(wg + (size_t)(j * 32 + pl) * p.Kpad + t * C1 + ks * 16 + g * 8);
    y5_wait_vm<0>();  // retired here, ahead of the counted-vmcnt ring: none of these loads may sit in the queue the tile loop counts
  }
  const int orow = lane / SPR, oslot = lane % SPR;
  constexpr bool STATS = !ACT && !RES && NT2 == 0;   // act = 0 instantiations: optional BatchNorm statistics of the train-mode forward (p.bn_partial)
  Y5StatAcc stat[1];
  if constexpr (STATS) stat[0].clear();
  const int orow2 = lane / SPR2, oslot2 = lane % SPR2;
  T* __restrict__ y2g = static_cast<T*>(p.y2);

  // ---- tile schedule ---------------------------------------------------------------------------------------------
  const int tw = p.OW / TC, th = p.OH / TR;
  const int nwt = p.B * th * tw;
  const int G = gridDim.x, bid = blockIdx.x;
  const int nbt = (nwt + NWV - 1) / NWV;
  const int nmine = (nbt - bid + G - 1) / G;
  auto tile_id = [&](int j) { return y5_xcd_remap(bid + j * G, nbt) * NWV + wave; };
  int nw = nmine;
  if (nw > 0 && tile_id(nw - 1) >= nwt) --nw;
  auto tile_origin = [&](int j, int& b, int& oh0, int& ow0) {
    const int t = tile_id(j);
    const int tx = t % tw, r = t / tw;
    const int ty = r % th;
    b = r / th; oh0 = ty * TR; ow0 = tx * TC;
  };

  auto issue = [&](int j, int buf) {
    int b, oh0, ow0;
    tile_origin(j, b, oh0, ow0);
    const int ih0 = oh0 * SH - 1, iw0 = ow0 * SH - 1;
    const int base = ((b * p.H + ih0) * p.W + iw0) * p.ldx * 2;
    const bool interior = ih0 >= 0 && ih0 + RH <= p.H && iw0 >= 0 && iw0 + RW <= p.W;  // wave-uniform
    char* dst = ring + buf * STAGE;
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      bool ok = (pc_rc[I] >> 16) != 0;
      if (!interior) {
        const int ih = ih0 + (pc_rc[I] & 0xff), iw = iw0 + ((pc_rc[I] >> 8) & 0xff);
        ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      }
      y5_bglds16(xrs, ok ? (unsigned)(base + pc_rel[I]) : Y5_OOB, dst + I * 1024);
    }
  };

  for (int s = 0; s < S; ++s)
    if (s < nw) issue(s, s);

#ifdef Y5_K3_TIMING
  unsigned long long d_wait = 0, d_res = 0, d_mfma = 0, d_epi = 0, d_issue = 0;
#endif
  int buf = 0;
  for (int i = 0; i < nw; ++i) {
#ifdef Y5_K3_TIMING
    const unsigned long long k_t0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int PER = LP + SP + RP;
    if (i + S - 1 >= nw) {
      y5_wait_vm<0>();
    } else if (i < S - 1) {
      switch (i) {
        case 0: y5_wait_vm<(S - 1) * LP>(); break;
        case 1: y5_wait_vm<(S - 1) * LP + (SP + RP)>(); break;
        case 2: y5_wait_vm<(S - 1) * LP + 2 * (SP + RP)>(); break;
        default: y5_wait_vm<(S - 1) * LP>(); break;
      }
    } else {
      y5_wait_vm<(S - 1) * PER>();
    }
    __builtin_amdgcn_wave_barrier();
#ifdef Y5_K3_TIMING
    const unsigned long long k_t1 = __builtin_amdgcn_s_memtime();
#endif
    char* st = ring + buf * STAGE;
    int b, oh0, ow0;
    tile_origin(i, b, oh0, ow0);
    // output pixel of this lane's store rows (row = ps*RPP + orow of the 32 tile pixels)
    uint4_t resv[NPASS];
    if constexpr (RES) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int row = ps * RPP + orow;
        const size_t m = ((size_t)b * p.OH + oh0 + (row >> 3)) * p.OW + ow0 + (row & 7);
        resv[ps] = *reinterpret_cast<const uint4_t*>(rg + m * p.ldr + oslot * 8);
        Y5_EMU_VM_OP(true);   // (a register load: one of the RP operations the counted waits skip)
      }
    }
#ifdef Y5_K3_TIMING
    const unsigned long long k_t2 = __builtin_amdgcn_s_memtime();
#endif
    // one accumulation chain per output block (a dependent MFMA takes its SrcC from the previous one without a bubble that would pay for the
    // 16 v_accvgpr_read + 8 v_pk_add_f32 per block a second accumulator costs in the epilogue, and for its 16 registers at two waves per SIMD)
    float16_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8_t af = *reinterpret_cast<const half8_t*>(st + rd[t][ks]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          half8_t wf;
          if constexpr (WREG) wf = wreg[t][ks][j];
          else wf = *reinterpret_cast<const half8_t*>(wlds + j * 32 * K2 + t * NSL * 16 + wsl[ks]);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc[j], 0, 0, 0);
        }
      }
#ifdef Y5_K3_TIMING
    asm volatile("" :: "v"(acc[0][0]));
    const unsigned long long k_t3 = __builtin_amdgcn_s_memtime();
#endif
    // epilogue: bias + SiLU -> scratch (vacated stage) -> (+ residual) -> 16-byte stores
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4_t bv = *reinterpret_cast<const float4_t*>(blds + j * 32 + q * 8 + g * 4);
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = acc[j][q * 4 + e] + bv[e]; o[e] = (half_t)(ACT ? y5_silu(t) : t); }
        const int slot = j * 4 + q;
        *reinterpret_cast<half4_t*>(st + pl * (NPAD * 2) + ((slot ^ (pl & SWM)) * 16) + g * 8) = o;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (NT2 > 0) {
      // ---- fused 1x1: the parked tile is the activation operand (lane = pixel row pl, k-half g), K = NPAD channels -----------------
      float16_t acc3[NT2];
#pragma unroll
      for (int j = 0; j < NT2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
      const int sw2 = NPAD >= 64 ? ((pl >> 1) & 7) : ((pl >> 2) & 3);
#pragma unroll
      for (int ks = 0; ks < NPAD / 16; ++ks) {
        const half8_t af = *reinterpret_cast<const half8_t*>(st + pl * (NPAD * 2) + (((ks * 2 + g) ^ (pl & SWM)) * 16));
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
          const half8_t wf = *reinterpret_cast<const half8_t*>(w2lds + (j * 32 + pl) * K2B + (((ks * 2 + g) ^ sw2) * 16));
          acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc3[j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every lane has read the parked tile: the stage becomes the second epilogue's scratch
#pragma unroll
      for (int j = 0; j < NT2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4_t bv = *reinterpret_cast<const float4_t*>(b2lds + j * 32 + q * 8 + g * 4);
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float t = acc3[j][q * 4 + e] + bv[e]; o[e] = (half_t)(p.pw2_act ? y5_silu(t) : t); }
          const int slot = j * 4 + q;
          *reinterpret_cast<half4_t*>(st + pl * (NPAD2 * 2) + ((slot ^ (pl & SWM2)) * 16) + g * 8) = o;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < NPASS2; ++ps) {
        const int row = ps * RPP2 + orow2;
        const uint4_t raw = *reinterpret_cast<const uint4_t*>(st + row * (NPAD2 * 2) + ((oslot2 ^ (row & SWM2)) * 16));
        const size_t m = ((size_t)b * p.OH + oh0 + (row >> 3)) * p.OW + ow0 + (row & 7);
        const int n = oslot2 * 8;
        if (n < p.pw2_c2) {
          T* d = n < p.pw2_split ? yg + m * p.ldy + n : y2g + m * p.ld2 + (n - p.pw2_split);  // ONE store instruction either way (counted vmcnt)
          *reinterpret_cast<uint4_t*>(d) = raw;
        }
        Y5_EMU_VM_OP(true);   // (lane slot 0 holds channel 0: the wave always issues it)
      }
    } else {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + orow;
      uint4_t raw = *reinterpret_cast<const uint4_t*>(st + row * (NPAD * 2) + ((oslot ^ (row & SWM)) * 16));
      if constexpr (RES) {
        half8_t a = __builtin_bit_cast(half8_t, raw), r8 = __builtin_bit_cast(half8_t, resv[ps]), c;
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] = (half_t)((float)a[e] + (float)r8[e]);
        raw = __builtin_bit_cast(uint4_t, c);
      }
      const size_t m = ((size_t)b * p.OH + oh0 + (row >> 3)) * p.OW + ow0 + (row & 7);
      const int n = oslot * 8;
      if constexpr (STATS) {
        if (p.bn_partial) stat[0].add(raw);   // (a wave tile is 4 x 8 real pixels: H % 4 == 0, W % 8 == 0)
      }
      if (n < p.C2) *reinterpret_cast<uint4_t*>(yg + m * p.ldy + n) = raw;
      Y5_EMU_VM_OP(true);
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef Y5_K3_TIMING
    const unsigned long long k_t4 = __builtin_amdgcn_s_memtime();
#endif
    if (i + S < nw) issue(i + S, buf);
    buf = buf + 1 == S ? 0 : buf + 1;
#ifdef Y5_K3_TIMING
    const unsigned long long k_t5 = __builtin_amdgcn_s_memtime();
    d_wait += k_t1 - k_t0; d_res += k_t2 - k_t1; d_mfma += k_t3 - k_t2; d_epi += k_t4 - k_t3; d_issue += k_t5 - k_t4;
#endif
  }
  if constexpr (STATS) {
    if (p.bn_partial) {   // (kernel-uniform) every wave's ring is idle: its first stage carries the wave's sums to the cross-wave addition
      y5_wait_vm<0>();
      y5_stat_flush<SPR, 1, NWV>(stat, lane, tid, ring, smem + W_BYTES + NPAD * 4 + NPAD2 * (K2B + 4), S * STAGE, p.bn_partial + (size_t)blockIdx.x * 2 * p.C2, p.C2);
    }
  }
#ifdef Y5_K3_TIMING
  if (lane == 0 && blockIdx.x == 0) {
    unsigned long long* o = y5_k3_dbg + wave * 8;
    o[0] = d_wait; o[1] = d_res; o[2] = d_mfma; o[3] = d_epi; o[4] = d_issue; o[5] = nw;
  }
#endif
}
