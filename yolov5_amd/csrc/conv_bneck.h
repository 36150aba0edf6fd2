// Fused Bottleneck for the HBM-bound high-resolution C3 blocks (models/common.py:164-181 with e = 1.0 inside C3, :242):
//     y = x + cv2(cv1(x)),  cv1 = 1x1 C->C + bias + SiLU,  cv2 = 3x3 pad 1 C->C + bias + SiLU      (BN folded, fp16, C = 32 / 64)
// in ONE pass: the 1x1 output `t` never reaches HBM.  Unfused, a Bottleneck at P2 of yolov5s bs=64 moves 105 MB (x) + 105 (t written)
// + 105 (t read) + 105 (x again, residual) + 105 (y) = 525 MB; fused 210 MB.
// Streaming structure of conv_k3.h (filters resident in LDS, every wave on its own, counted vmcnt, no barriers in the loop):
//   * a wave tile is a 4 x 8 block of output pixels; its 6 x 10 receptive field of x (all C channels) is staged once by LDS-DMA
//     (whole pixels, 16-byte pieces XOR-swizzled on the source side);
//   * GEMM 1: t = SiLU(W1 x + b1) for the 60 receptive-field pixels (two 32-row MFMA tiles), written back to LDS in the same pixel-row
//     layout -- ZERO for pixels outside the image (the 3x3's padding applies to t, not to x);
//   * the residual (the tile's own 32 pixels of x) is lifted from the stage into registers, after which the stage is dead and the
//     receptive field of tile i + S is requested into it (S stages per wave: the request has S - 1 whole tiles + GEMM 2 to land);
//   * GEMM 2: nine taps = nine LDS row offsets into t (exactly conv_k3.h's fragment addressing); epilogue bias + SiLU, transposed
//     through the t buffer (free by then), + residual, 16-byte row-contiguous stores.
// vmcnt (loads and stores retire in issue order on one counter): the prologue issues stages 0 .. S-1; tile j issues stage j + S (LP loads)
// after its GEMM 1 and its SP stores at its end.  Behind stage i's loads the queue therefore holds, at the top of tile i,
// (S - 1 - i) * LP + i * (LP + SP) operations while the ring fills (i < S) and SP + (S - 1) * (LP + SP) in steady state; in the last
// S - 1 tiles (nothing left to request) the wave drains completely.
#pragma once
#include "conv_pw.h"

// kernel-experiment builds only (scripts/bneck_ablate.sh): -DY5_BNECK_ABL=<bits> 1 = no SiLU, 2 = no GEMM-2 MFMAs, 4 = stage only the
// first tile, 8 = no global stores, 16 = no GEMM-1 (t = 0)
#ifndef Y5_BNECK_ABL
#define Y5_BNECK_ABL 0
#endif
__device__ __forceinline__ float y5_bneck_act(float v) { return (Y5_BNECK_ABL & 1) ? v : y5_silu(v); }

template <int C, int S, bool CV3 = false, int NWV = 4, bool ALIAS = false>
constexpr size_t y5_conv_bneck_lds_bytes() {
  constexpr int NSL = C / 8, NI = (60 * NSL + 63) / 64;
  // CV3: + the third bias (the third filter lives in registers, the cv2 half of its input comes straight from global memory)
  // ALIAS: the t buffer IS the (single) stage -- see the kernel
  return (size_t)C * C * 2 + (size_t)C * 9 * C * 2 + (size_t)2 * C * 4 +
         (size_t)NWV * (ALIAS ? (NI * 1024 > 64 * C * 2 ? NI * 1024 : 64 * C * 2) : S * NI * 1024 + 64 * C * 2) + (CV3 ? (size_t)2 * C * 4 : 0);
}

struct Y5BneckParams {
  const void* x;       // NHWC slice, pixel stride ldx
  const void* w1;      // packed 1x1 filter [C][Kpad1]
  const void* w2;      // packed 3x3 filter [C][Kpad2], k = (kh, kw, c)
  const float* b1;
  const float* b2;
  void* y;             // NHWC slice, pixel stride ldy (may alias x: every tile reads its receptive field before any neighbour... NO: must not alias)
  unsigned x_bytes, w1_bytes, w2_bytes;
  int B, H, W, ldx, ldy, Kpad1, Kpad2, add;
  // CV3 kernels: C3's cv3 (1x1 over cat(m(cv1(x)), cv2(x)), models/common.py:246) applied to the finished tile: the other half of its input,
  // the third filter [2C padded][Kpad3] (k = (this kernel's C outputs, then y2's C channels)), bias, activation, real output channels;
  // the result goes to y (pixel stride ldy) INSTEAD of the Bottleneck's own output
  const void* y2;
  const void* w3;
  const float* b3;
  unsigned y2_bytes, w3_bytes;
  int ld2, Kpad3, C3, act3;
};

// NWV = 8 + ALIAS (round 4, C = 64): eight waves share the one copy of the filters (8 + 72 KB at C = 64) and each wave's t buffer IS its single
// stage -- GEMM 1 reads every fragment of the receptive field and the residual is lifted to registers BEFORE t is written over it, and the next tile's
// receptive field is requested only when the tile's last scratch read has returned (no prefetch inside a wave: the other wave of the SIMD covers the
// round trip, the "resident waves beat prefetch depth" finding of conv_k3.h / conv_pw.h).  Four waves with stage + t apart fill the LDS at C = 64.
template <int C, int S, bool ADD, bool CV3 = false, int NWV = 4, bool ALIAS = false>
__global__ __launch_bounds__(NWV * 64, CV3 ? 3 : 1)  // CV3: three waves per SIMD as without it (the third filter's 32 registers must not cost a workgroup per CU)
void y5_conv_bneck_kernel(const Y5BneckParams p) {
  static_assert(!ALIAS || (S == 1 && !CV3), "the aliased form has one stage and no third GEMM");
  typedef half_t T;
  constexpr int NT = C / 32;
  constexpr int TR = 4, TC = 8, RH = TR + 2, RW = TC + 2;   // wave tile / receptive field
  constexpr int NSL = C / 8, ROWB = C * 2;
  constexpr int NPIX = RH * RW, NPIECE = NPIX * NSL, NI = (NPIECE + 63) / 64, STAGE = NI * 1024;
  constexpr int TBYTES = 64 * ROWB;                          // t buffer: two 32-row MFMA tiles
  constexpr int KS = C / 16;                                 // MFMA k-steps per tap / of GEMM 1
  constexpr int K1B = C * 2, K2B = 9 * C * 2;                // bytes per filter row in LDS
  constexpr int W1_BYTES = C * K1B, W2_BYTES = C * K2B;
  constexpr int SPR = C / 8, RPP = 64 / SPR, NPASS = 32 / RPP, SWM = SPR >= 8 ? 7 : SPR - 1;
  constexpr int C3P = 2 * C;                                 // CV3: (padded) output channels of the third GEMM = its K
  constexpr int STAGE2 = STAGE;
  constexpr int SPR3 = C3P / 8, RPP3 = 64 / SPR3, NPASS3 = 32 / RPP3, SWM3 = SPR3 >= 8 ? 7 : SPR3 - 1;
  constexpr int SP = CV3 ? NPASS3 : NPASS, LP = NI;          // stores / LDS-DMA loads per tile per wave (CV3's two register loads of the cv2 half are
                                                             // issued at the top of their tile: older than everything the counted waits leave in flight)
  static_assert(!CV3 || (C == 32 && S == 1 && TBYTES >= 32 * C3P * 2), "cv3 fusion is built for C = 32 (third GEMM 64 -> 64), one ring stage");
  static_assert(S >= 1 && S <= 3, "1 to 3 stages");
  static_assert(TBYTES >= 32 * C * 2, "epilogue scratch must fit in the t buffer");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w1l = smem;
  char* w2l = smem + W1_BYTES;
  float* b1l = reinterpret_cast<float*>(smem + W1_BYTES + W2_BYTES);
  float* b2l = b1l + C;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* b3l = reinterpret_cast<float*>(smem + W1_BYTES + W2_BYTES + 2 * C * 4);
  constexpr int PERWAVE = ALIAS ? (STAGE2 > TBYTES ? STAGE2 : TBYTES) : S * STAGE2 + TBYTES;
  char* ring = smem + W1_BYTES + W2_BYTES + 2 * C * 4 + (CV3 ? C3P * 4 : 0) + wave * PERWAVE;
  char* ts = ALIAS ? ring : ring + S * STAGE2;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t w1rs = y5_make_rsrc(p.w1, p.w1_bytes);
  const y5_rsrc_t w2rs = y5_make_rsrc(p.w2, p.w2_bytes);
  const y5_rsrc_t y2rs = y5_make_rsrc(CV3 ? p.y2 : p.x, CV3 ? p.y2_bytes : 0u);
  T* __restrict__ yg = static_cast<T*>(p.y);

  auto fsw = [](int q) { return C == 32 ? ((q >> 2) & 3) : ((q >> 1) & 7); };  // pixel-row swizzle (64 / 128-byte rows)
  // The t buffer has its own swizzle (round 6).  GEMM 2's fragment rows are NOT consecutive -- lane pl reads pixel (pl >> 3) * 10 + (pl & 7) + tap -- and under
  // fsw every one of its ds_read_b128 was 3-way bank-conflicted (the "36-43 % of LDS-active cycles" of the round-5 counters; modelled lane group by lane group
  // in DESIGN.md section 4.8).  No function of the pixel row is conflict-free for all nine taps at row pitch 10 (pitch 16 is, and does not fit the LDS at C = 64);
  // these two bring the reads from 3.0 to 1.9 LDS cycles per lane group and leave the t stores at their 2.0.
#ifdef Y5_BNECK_OLD_TSW   // (A/B build of profiles/r06/r06_ab_bneck_tsw.log)
  auto tsw = fsw;
#else
  auto tsw = [](int q) { return C == 32 ? (((q >> 2) + 3 * (q >> 4)) & 3) : (((q >> 1) + 6 * (q >> 4)) & 7); };
#endif

  // ---- prologue: both filters (rows swizzled like the activation rows) + biases into LDS -----------------------------
  {
    constexpr int WI1 = C * NSL / 64;
    for (int I = wave; I < WI1; I += NWV) {
      const int pidx = I * 64 + lane;
      const int n = pidx / NSL, ps = pidx - n * NSL;
      y5_bglds16(w1rs, (unsigned)(n * p.Kpad1 * 2 + ((ps ^ fsw(n)) * 16)), w1l + I * 1024);
    }
    constexpr int WSL = 9 * NSL, WI2 = C * WSL / 64;
    for (int I = wave; I < WI2; I += NWV) {
      const int pidx = I * 64 + lane;
      const int n = pidx / WSL, ps = pidx - n * WSL;
      const int src_slot = (ps & ~(NSL - 1)) | ((ps & (NSL - 1)) ^ fsw(n));
      y5_bglds16(w2rs, (unsigned)(n * p.Kpad2 * 2 + src_slot * 16), w2l + I * 1024);
    }
    for (int i = tid; i < C; i += NWV * 64) { b1l[i] = p.b1[i]; b2l[i] = p.b2[i]; }
    if constexpr (CV3) {
      for (int i = tid; i < C3P; i += NWV * 64) b3l[i] = p.b3[i];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }

  // ---- per-lane constants ----------------------------------------------------------------------------------------------
  int pc_rel[NI], pc_rc[NI];  // staged pieces: byte offset relative to the tile's (ih0, iw0) pixel / r | cq << 8 | valid << 16
#pragma unroll
  for (int I = 0; I < NI; ++I) {
    const int idx = I * 64 + lane;
    const int q = idx / NSL, cs = idx - q * NSL;
    const int r = q / RW, cq = q - r * RW;
    pc_rel[I] = ((r * p.W + cq) * p.ldx) * 2 + ((cs ^ fsw(q)) * 16);
    pc_rc[I] = r | (cq << 8) | ((idx < NPIECE ? 1 : 0) << 16);
  }
  const int g = lane >> 5, pl = lane & 31;
  // GEMM 1: this lane's two receptive-field pixels (rows pl and 32 + pl of t) and their k-slot offsets in the stage
  int q1r[2], q1c[2], a1rd[2][KS];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int q = f * 32 + pl;
    q1r[f] = q / RW; q1c[f] = q - q1r[f] * RW;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a1rd[f][ks] = q * ROWB + (((ks * 2 + g) ^ fsw(q)) * 16);
  }
  int w1sl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) w1sl[ks] = pl * K1B + (((ks * 2 + g) ^ fsw(pl)) * 16);
  // GEMM 2: fragment addresses into t for tap (kh, kw), as conv_k3.h
  const int q0 = (pl >> 3) * RW + (pl & 7);
  int rd[9][KS];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int q = q0 + (t / 3) * RW + (t % 3);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rd[t][ks] = q * ROWB + (((ks * 2 + g) ^ tsw(q)) * 16);
  }
  int w2sl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) w2sl[ks] = pl * K2B + (((ks * 2 + g) ^ fsw(pl)) * 16);
  const int orow = lane / SPR, oslot = lane % SPR;
  // CV3: the third filter's fragments stay in registers for the lifetime of the workgroup (lane (n = pl, g), k-step ks: 8 halfs of row j*32 + pl)
  half8_t w3f[CV3 ? C3P / 32 : 1][CV3 ? 2 * KS : 1];
  if constexpr (CV3) {
    const T* w3g = static_cast<const T*>(p.w3);
#pragma unroll
    for (int j = 0; j < C3P / 32; ++j)
#pragma unroll
      for (int ks = 0; ks < 2 * KS; ++ks) w3f[j][ks] = *reinterpret_cast<const half8_t*>(w3g + (size_t)(j * 32 + pl) * p.Kpad3 + (ks * 2 + g) * 8);
  }

  // ---- tile schedule (conv_k3.h) -----------------------------------------------------------------------------------------
  const int tw = p.W / TC, th = p.H / TR;
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
    char* xs = ring + buf * STAGE2;
    int b, oh0, ow0;
    tile_origin(j, b, oh0, ow0);
    const int ih0 = oh0 - 1, iw0 = ow0 - 1;
    const int base = ((b * p.H + ih0) * p.W + iw0) * p.ldx * 2;
    const bool interior = ih0 >= 0 && ih0 + RH <= p.H && iw0 >= 0 && iw0 + RW <= p.W;  // wave-uniform
#pragma unroll
    for (int I = 0; I < NI; ++I) {
      bool ok = (pc_rc[I] >> 16) != 0;
      if (!interior) {
        const int ih = ih0 + (pc_rc[I] & 0xff), iw = iw0 + ((pc_rc[I] >> 8) & 0xff);
        ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      }
      y5_bglds16(xrs, ok ? (unsigned)(base + pc_rel[I]) : Y5_OOB, xs + I * 1024);
    }
  };

  for (int s0 = 0; s0 < S; ++s0)
    if (s0 < nw) issue(s0, s0);
  int buf = 0;
  for (int i = 0; i < nw; ++i) {
    if (ALIAS || i + S - 1 >= nw) {
      y5_wait_vm<0>();   // ALIAS: the stage's loads are the NEWEST operations of the queue (issued behind the previous tile's stores)
    } else if (i < S) {
      switch (i) {
        case 0: y5_wait_vm<(S - 1) * LP>(); break;
        case 1: y5_wait_vm<(S >= 2 ? (S - 2) * LP : 0) + (LP + SP)>(); break;
        default: y5_wait_vm<2 * (LP + SP)>(); break;  // i == 2, S == 3
      }
    } else {
      y5_wait_vm<SP + (S - 1) * (LP + SP)>();
    }
    __builtin_amdgcn_wave_barrier();
    char* xs = ring + buf * STAGE2;
    int b, oh0, ow0;
    tile_origin(i, b, oh0, ow0);
    // CV3: this lane's activation fragments of C3's cv2 half (pixel pl, channels (2 ks + g) * 8 .. + 7), straight into registers; they are
    // needed by the third GEMM only -- two whole GEMMs later
    uint4_t y2f[CV3 ? KS : 1];
    if constexpr (CV3) {
      const int poff = (((b * p.H + oh0 + (pl >> 3)) * p.W + ow0 + (pl & 7)) * p.ld2) * 2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) y2f[ks] = y5_buffer_load16(y2rs, poff + (ks * 2 + g) * 16, 0);
    }
    uint4_t resv[NPASS];
    // ---- GEMM 1: t = SiLU(W1 x + b1) on the receptive field, zero outside the image -------------------------------------
    {
      float16_t acc1[2][NT];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc1[f][j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < ((Y5_BNECK_ABL & 16) ? 0 : KS); ++ks) {
        half8_t af[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) af[f] = *reinterpret_cast<const half8_t*>(xs + a1rd[f][ks]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const half8_t wf = *reinterpret_cast<const half8_t*>(w1l + j * 32 * K1B + w1sl[ks]);
#pragma unroll
          for (int f = 0; f < 2; ++f) acc1[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af[f], acc1[f][j], 0, 0, 0);
        }
      }
      if constexpr (ALIAS && ADD) {   // t is about to overwrite the stage: the tile's own pixels of x go to registers first
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int row = ps * RPP + orow;
          const int q = ((row >> 3) + 1) * RW + (row & 7) + 1;
          resv[ps] = *reinterpret_cast<const uint4_t*>(xs + q * ROWB + ((oslot ^ fsw(q)) * 16));
        }
      }
      if constexpr (ALIAS) {
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): every fragment / residual read of the stage has returned
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int q = f * 32 + pl;
        const int ih = oh0 - 1 + q1r[f], iw = ow0 - 1 + q1c[f];
        const bool in_img = q < NPIX && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const float4_t bv = *reinterpret_cast<const float4_t*>(b1l + j * 32 + qq * 8 + g * 4);
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = in_img ? (half_t)y5_bneck_act(acc1[f][j][qq * 4 + e] + bv[e]) : (half_t)0.f;
            const int slot = j * 4 + qq;  // 16-byte slot = channel / 8
            *reinterpret_cast<half4_t*>(ts + q * ROWB + ((slot ^ tsw(q)) * 16) + g * 8) = o;
          }
      }
    }
    // ---- residual: the tile's own pixels of x, stage -> registers; then the stage is free for the next tile --------------
    if constexpr (ADD && !ALIAS) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int row = ps * RPP + orow;
        const int q = ((row >> 3) + 1) * RW + (row & 7) + 1;
        resv[ps] = *reinterpret_cast<const uint4_t*>(xs + q * ROWB + ((oslot ^ fsw(q)) * 16));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // t is written, every LDS read of the stage has returned (lgkmcnt 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    if constexpr (!ALIAS) {
      if (i + S < nw && !(Y5_BNECK_ABL & 4)) issue(i + S, buf);
      buf = buf + 1 == S ? 0 : buf + 1;
    }
    // ---- GEMM 2: 3x3 over t ---------------------------------------------------------------------------------------------------
    float16_t acc[NT];  // one accumulation chain per block, as conv_k3.h
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8_t af = *reinterpret_cast<const half8_t*>(ts + rd[t][ks]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const half8_t wf = *reinterpret_cast<const half8_t*>(w2l + j * 32 * K2B + t * NSL * 16 + w2sl[ks]);
          if (Y5_BNECK_ABL & 2) { asm volatile("" ::"v"(wf), "v"(af)); continue; }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc[j], 0, 0, 0);
        }
      }
    // ---- epilogue: bias + SiLU -> scratch (the t buffer) -> + residual -> 16-byte stores -----------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every lane's fragment reads of t are done before t is overwritten
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4_t bv = *reinterpret_cast<const float4_t*>(b2l + j * 32 + qq * 8 + g * 4);
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)y5_bneck_act(acc[j][qq * 4 + e] + bv[e]);
        const int slot = j * 4 + qq;
        *reinterpret_cast<half4_t*>(ts + pl * (C * 2) + ((slot ^ (pl & SWM)) * 16) + g * 8) = o;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (CV3) {
      // ---- cv3: the finished tile (+ residual) goes back into the scratch, and is the first half of the third GEMM's K; the second half are the
      // lifted fragments of C3's cv2 output.  K order = (m channels, cv2 channels) = the order of torch.cat at common.py:246.
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int row = ps * RPP + orow;
        char* q = ts + row * (C * 2) + ((oslot ^ (row & SWM)) * 16);
        uint4_t raw = *reinterpret_cast<const uint4_t*>(q);
        if constexpr (ADD) {
          half8_t a = __builtin_bit_cast(half8_t, raw), r8 = __builtin_bit_cast(half8_t, resv[ps]), c;
#pragma unroll
          for (int e = 0; e < 8; ++e) c[e] = (half_t)((float)a[e] + (float)r8[e]);
          raw = __builtin_bit_cast(uint4_t, c);
          *reinterpret_cast<uint4_t*>(q) = raw;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      constexpr int NT3 = C3P / 32;
      float16_t acc3[NT3];
#pragma unroll
      for (int j = 0; j < NT3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2 * KS; ++ks) {
        half8_t af;
        if (ks < KS) af = *reinterpret_cast<const half8_t*>(ts + pl * (C * 2) + (((ks * 2 + g) ^ (pl & SWM)) * 16));
        else af = __builtin_bit_cast(half8_t, y2f[ks - KS]);
#pragma unroll
        for (int j = 0; j < NT3; ++j) acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[j][ks], af, acc3[j], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every lane has read the tile: the scratch takes the third epilogue
#pragma unroll
      for (int j = 0; j < NT3; ++j)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float4_t bv = *reinterpret_cast<const float4_t*>(b3l + j * 32 + qq * 8 + g * 4);
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float t = acc3[j][qq * 4 + e] + bv[e]; o[e] = (half_t)(p.act3 ? y5_bneck_act(t) : t); }
          const int slot = j * 4 + qq;
          *reinterpret_cast<half4_t*>(ts + pl * (C3P * 2) + ((slot ^ (pl & SWM3)) * 16) + g * 8) = o;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int orow3 = lane / SPR3, oslot3 = lane % SPR3;
#pragma unroll
      for (int ps = 0; ps < NPASS3; ++ps) {
        const int row = ps * RPP3 + orow3;
        const uint4_t raw = *reinterpret_cast<const uint4_t*>(ts + row * (C3P * 2) + ((oslot3 ^ (row & SWM3)) * 16));
        const size_t m = ((size_t)b * p.H + oh0 + (row >> 3)) * p.W + ow0 + (row & 7);
        const int n = oslot3 * 8;
        if (n < p.C3) *reinterpret_cast<uint4_t*>(yg + m * p.ldy + n) = raw;
        Y5_EMU_VM_OP(true);   // (lane slot 0 holds channel 0: the wave always issues it)
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int row = ps * RPP + orow;
        uint4_t raw = *reinterpret_cast<const uint4_t*>(ts + row * (C * 2) + ((oslot ^ (row & SWM)) * 16));
        if constexpr (ADD) {
          half8_t a = __builtin_bit_cast(half8_t, raw), r8 = __builtin_bit_cast(half8_t, resv[ps]), c;
  #pragma unroll
          for (int e = 0; e < 8; ++e) c[e] = (half_t)((float)a[e] + (float)r8[e]);
          raw = __builtin_bit_cast(uint4_t, c);
        }
        const size_t m = ((size_t)b * p.H + oh0 + (row >> 3)) * p.W + ow0 + (row & 7);
        if (!(Y5_BNECK_ABL & 8)) *reinterpret_cast<uint4_t*>(yg + m * p.ldy + oslot * 8) = raw;
        else asm volatile("" ::"v"(raw));
        Y5_EMU_VM_OP(true);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the scratch is rewritten by the next tile's GEMM 1
    if constexpr (ALIAS) {
      __builtin_amdgcn_s_waitcnt(0xC07F);   // the last scratch reads have returned: the buffer takes the next receptive field
      __builtin_amdgcn_wave_barrier();
      if (i + 1 < nw && !(Y5_BNECK_ABL & 4)) issue(i + 1, 0);
    }
  }
}
