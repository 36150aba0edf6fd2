// 3x3 (pad 1) convolution with the input HALO resident in LDS -- the deep Bottleneck.cv2 layers of the backbone / neck
// (models/common.py:164-181 `Bottleneck`: cv2 = Conv(c_, c2, 3, 1); yolov5s: 64->64 @80^2, 128->128 @40^2, 256->256 @20^2).
//
// Why a second 3x3 kernel: the general implicit GEMM (conv_igemm.h) stages an im2col chunk per K step, so every input element
// crosses L2 -> LDS nine times and a 128x128 tile needs 16 LDS-DMA pieces per 16 MFMAs; its main loop is bound by that piece rate,
// not by the matrix cores.  Here a workgroup owns a SPATIAL tile of TH x TW output pixels of one image:
//   * per 32-channel chunk the (TH+2) x (TW+2) input halo is staged ONCE (64 B per pixel, 16-byte slots XOR-swizzled on the source
//     side, out-of-image pixels = out-of-range buffer offsets = zero fill) and serves all nine taps: the MFMA activation fragment of
//     lane (pixel m, k-half g) for tap (kh, kw) is one ds_read_b128 at halo pixel hp0(m) + kh * (TW+2) + kw;
//   * only the filter streams per tap: BN rows x 64 B per (tap, chunk) step through a NINE-stage ring (stage = tap), issued EIGHT steps
//     ahead -- measured on MI355X a 3-stage ring (two steps = 0.6 us ahead) left the loop waiting on L2 latency: what bounds these
//     kernels is the number of bytes in flight per CU, so the ring is as deep as LDS allows (64 KB of filter + the next halo in flight);
//   * retired with COUNTED vmcnt: every wave issues the same, statically known number of LDS-DMA instructions per step (halo pieces a
//     wave does not own go to a per-wave dummy slot with an out-of-range source: no traffic), so the wait immediate of step t is a
//     compile-time constant; one raw s_barrier per step;
//   * the next chunk's halo is issued in the first two steps of the current chunk (seven steps ahead of its first use).
// L2 -> LDS traffic per MFMA drops ~6x against the im2col form (BM = 320: 3 pieces per wave per 20 MFMAs instead of 16 per 16).
// Tile geometry (TH, TW) is a launch parameter chosen by the host so that the tile count fills the CUs in whole rounds.
#pragma once
#include "conv_igemm.h"

#ifdef Y5_H3_TIMING
__device__ unsigned long long y5_h3_dbg[64];      // workgroup 0, per wave: wait+barrier, issue, compute (s_memtime ticks), steps
__device__ unsigned long long y5_h3_blocks[4096];  // per workgroup: kernel entry, first step, loop end, exit (s_memrealtime, 100 MHz)
#endif

template <int WM, int WN, int TM, int TN, int HPMAX, int NSW_ = 9>
struct Y5H3Geom {
  static constexpr int NW = WM * WN;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static constexpr int NAI_MAX = (HPMAX + 15) / 16;  // LDS-DMA instructions per activation stage (16 halo pixels x 64 B each)
  static constexpr int A_STAGE = NAI_MAX * 1024;
  static constexpr int W_INSTR = BN / 16;            // per filter stage: BN rows x 64 B
  static constexpr int WPW = (W_INSTR + NW - 1) / NW;  // filter pieces per wave per step (a wave without a piece issues a dummy: uniform counts)
  static constexpr int W_STAGE = BN * 64;
  static constexpr int NSW = NSW_;                     // filter ring: 9 = one stage per tap, slices issued eight steps ahead; 4 = three ahead,
                                                       // 32 KB instead of 72 KB so that two workgroups share a CU
  static constexpr int WIN = NSW - 3;                  // steps whose LDS-DMA instructions may still be in flight at a step's barrier
  static constexpr int APS = (NAI_MAX + NW - 1) / NW;  // activation pieces per wave per chunk
  static constexpr int PPS = (APS + 1) / 2;            // ... issued in the first two steps of the previous chunk (seven steps ahead)
  static constexpr int SCR_ROWB = 32 * 2 + 16, SCR_BYTES = 32 * SCR_ROWB;
  // the epilogue's transposition scratch lives in halo stage 1 (idle between a tile's last step and the next tile's step 0)
  static constexpr int BIAS_MAX = 1024;               // output channels whose fp32 bias is staged in LDS (4 KB) once per workgroup
  static constexpr size_t OFF_BIAS = (size_t)2 * A_STAGE + (size_t)NSW * W_STAGE + 1024;  // behind one dummy slot (zeros from any wave)
  static constexpr size_t LDS = OFF_BIAS + (size_t)BIAS_MAX * 4;
  static_assert(NW * SCR_BYTES <= A_STAGE, "epilogue scratch must fit into a halo stage");
  static_assert(NSW == 9 || NSW == 4, "ring depths with bookkeeping: 9 and 4");
  // LDS-DMA instructions (dummies included) a wave has issued for the NEXT chunk's halo (taps 0 and 1) within the WIN steps before tap t
  static constexpr int cur_a(int t) { return (t - WIN <= 0 && 0 <= t - 1 ? PPS : 0) + (t - WIN <= 1 && 1 <= t - 1 ? APS - PPS : 0); }
  // inside the LAST chunk a step x issues a filter slice only while one NSW-1 steps ahead exists (x <= 9 - NSW); steps of the chunk
  // before (x < 0) always did: number of filter-issuing steps among the WIN steps before tap t
  static constexpr int last_w(int t) {
    int n = 0;
    for (int x = t - WIN; x <= t - 1; ++x) n += x <= 9 - NSW ? 1 : 0;
    return n;
  }
};

template <int N> __device__ __forceinline__ void y5_wait_vm_dyn(int n) {  // n is wave-uniform, 0 <= n <= N
  if constexpr (N == 0) {
    y5_wait_vm<0>();
  } else {
    if (n >= N) y5_wait_vm<N>();
    else y5_wait_vm_dyn<N - 1>(n);
  }
}

template <int WM, int WN, int TM, int TN, int HPMAX, int NSW_ = 9>
__global__ __launch_bounds__(WM * WN * 64, NSW_ == 4 ? 2 : 1)
void y5_conv_h3_kernel(const Y5ConvParams p) {
  typedef half_t T;
  using Gm = Y5H3Geom<WM, WN, TM, TN, HPMAX, NSW_>;
  constexpr int NW = Gm::NW, BN = Gm::BN;
  constexpr int A_STAGE = Gm::A_STAGE, W_STAGE = Gm::W_STAGE, WPW = Gm::WPW, APS = Gm::APS, NSW = Gm::NSW;
  constexpr int SCR_ROWB = Gm::SCR_ROWB;

#ifdef Y5_H3_TIMING
  const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
  unsigned long long r_first = 0, r_loop_end = 0, d_wait = 0, d_issue = 0, d_comp = 0, d_steps = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const a_lds = smem;                 // two halo stages
  char* const w_lds = smem + 2 * A_STAGE;   // three filter stages

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 5, frow = lane & 31;
  char* const scratch = smem + A_STAGE + wave * Gm::SCR_BYTES;  // halo stage 1
  char* const dummy = smem + 2 * A_STAGE + NSW * W_STAGE;  // shared by all waves: only ever receives zero fill, never read
  // fp32 bias of every (padded) output channel, staged once: the epilogue reads it with one ds_read_b128 per four channels.  (Round 5: the
  // `g ? pb[q*8+4+e] : pb[q*8+e]` form meant for the scalar cache had been compiled into 16 SERIALIZED per-lane global loads per accumulator
  // block, each behind an s_waitcnt vmcnt(0) that also drained the next tile's in-flight LDS-DMA.)
  float* const bias_lds = reinterpret_cast<float*>(smem + Gm::OFF_BIAS);
  for (int i = tid; i < p.Npad; i += NW * 64) bias_lds[i] = p.bias[i];

  // stride 1 or 2 (round 5: the 3x3 s2 down-sampling layers -- 3 / 5 / 7 / 18 / 21.Conv -- through the same kernel): output pixel (r, c) of the tile reads
  // halo pixel (r S + kh, c S + kw) of a ((TH-1) S + 3) x ((TW-1) S + 3) input halo; only the fragment base addresses and the halo extent know about S.
  // The input crosses L2 -> LDS ((TH-1) S + 3)((TW-1) S + 3) / (S^2 TH TW) = 1.1-1.2x instead of the implicit GEMM's 9 / S^2 = 2.25x.
  const int SD = p.SH;
  const int TH = p.h3_th, TW = p.h3_tw, HW = (TW - 1) * SD + 3;
  const int HP = ((TH - 1) * SD + 3) * HW;
  const int NAI = (HP + 15) >> 4;
  const int NCC = p.C1 >> 5;
  const int total = NCC * 9;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t wrs = y5_make_rsrc(p.w, p.w_bytes);

  // ---- per-lane constants ------------------------------------------------------------------------------------------------------
  // (a) the halo pieces this wave stages: instruction I = k * NW + wave covers 16 halo pixels x 4 slots
  int a_rel[APS], a_rc[APS];
#pragma unroll
  for (int k = 0; k < APS; ++k) {
    const int idx = (k * NW + wave) * 64 + lane;
    const int hp = idx >> 2, ds = idx & 3;
    const int ss = ds ^ ((hp >> 2) & 3);
    const int hr = hp / HW, hc = hp - hr * HW;
    a_rel[k] = ((hr * p.W + hc) * p.ldx) * 2 + ss * 16;
    a_rc[k] = hr | (hc << 8) | ((hp < HP ? 1 : 0) << 16);
  }
  // (b) fragment reads: lane (pixel row frow of block i, k-half g) -> halo pixel of tap (0,0); filter row of block j
  int hp0[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = (wm * TM + i) * 32 + frow;
    const int r = m / TW, c = m - r * TW;
    hp0[i] = r < TH ? r * SD * HW + c * SD : 0;
  }
  const int fsw = (frow >> 2) & 3;
  int w_rd[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) w_rd[j] = ((wn * TN + j) * 32 + frow) * 64 + ((g ^ fsw) << 4);
  // (c) store rows of the epilogue: lane -> tile pixel (pass ps, row ps*16 + lane/4), 16-byte vector lane%4 of the 32 channels
  int o_rc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = (wm * TM + i) * 32 + ps * 16 + (lane >> 2);
      const int r = m / TW, c = m - r * TW;
      o_rc[i][ps] = r < TH ? (r | (c << 16)) : -1;
    }

  // ---- tile schedule -----------------------------------------------------------------------------------------------------------
  const int G = gridDim.x, bid = blockIdx.x;
  const int tiles_s = p.B * p.h3_tiles_h * p.h3_tiles_w;
  const int ntiles = tiles_s * p.tilesN;
  const int nmine = (ntiles - bid + G - 1) / G;
  auto tile_coords = [&](int j, int& b, int& oh0, int& ow0, int& n0) {
    const int t = y5_xcd_remap(bid + j * G, ntiles);
    const int tn = t % p.tilesN, ts = t / p.tilesN;
    const int tx = ts % p.h3_tiles_w, q = ts / p.h3_tiles_w;
    const int ty = q % p.h3_tiles_h;
    b = q / p.h3_tiles_h;
    oh0 = ty * TH;
    ow0 = tx * TW;
    n0 = tn * BN;
  };

  // ---- loader state of the tile being staged -----------------------------------------------------------------------------------
  int s_base = 0, s_ih0 = 0, s_iw0 = 0;
  unsigned w_off[WPW];
  auto loader_setup = [&](int j) {
    int b, oh0, ow0, n0;
    tile_coords(j, b, oh0, ow0, n0);
    s_ih0 = oh0 * SD - 1;
    s_iw0 = ow0 * SD - 1;
    s_base = ((b * p.H + s_ih0) * p.W + s_iw0) * p.ldx * 2;
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
      const int row = (q * NW + wave) * 16 + (lane >> 2);
      const int ss = (lane & 3) ^ ((row >> 2) & 3);
      const int n = n0 + row;
      w_off[q] = n < p.Npad ? (unsigned)((n * p.Kpad) * 2 + ss * 16) : 0x80000000u;
    }
  };
  auto issue_a = [&](auto kc, int cc, bool with_dummy) {  // piece k of chunk cc; a wave without that piece issues a dummy (uniform counts)
    constexpr int k = decltype(kc)::value;
    const int I = k * NW + wave;
    if (I < NAI) {
      const int ih = s_ih0 + (a_rc[k] & 0xff), iw = s_iw0 + ((a_rc[k] >> 8) & 0xff);
      const bool ok = (a_rc[k] >> 16) != 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      y5_bglds16(xrs, ok ? (unsigned)(s_base + a_rel[k] + cc * 64) : Y5_OOB, a_lds + (cc & 1) * A_STAGE + I * 1024);
    } else if (with_dummy) {
      y5_bglds16_dummy(xrs, dummy);
    }
  };
  auto issue_w = [&](int stage, int tap, int cc) {
    const unsigned koff = (unsigned)((tap * p.C1 + cc * 32) * 2);
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
      if (Gm::W_INSTR % NW == 0 || q * NW + wave < Gm::W_INSTR) y5_bglds16(wrs, w_off[q] + koff, w_lds + stage * W_STAGE + (q * NW + wave) * 1024);
      else y5_bglds16_dummy(wrs, dummy);
    }
  };
  auto prologue = [&](int j) {  // first halo chunk + the filter slices of steps 0..NSW-2 (a tile has at least nine steps)
    loader_setup(j);
    y5_static_for<0, APS>([&](auto kc) { issue_a(kc, 0, false); });
    y5_static_for<0, NSW - 1>([&](auto uc) { issue_w(decltype(uc)::value, decltype(uc)::value, 0); });
  };

  float16_t acc[TM][TN];

  // ---- epilogue: bias + SiLU -> per-wave LDS transpose -> (+ residual) -> 16-byte row-contiguous stores ----------------------------
  T* yg = static_cast<T*>(p.y);  // may alias p.res (in-place residual)
  const T* rg = static_cast<const T*>(p.res);
  auto epilogue = [&](int b, int oh0, int ow0, int n0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nt = n0 + (wn * TN + j) * 32;
        const float* pb = bias_lds + (nt < p.Npad ? nt : 0) + g * 4;
        const float keep = nt < p.Npad ? 1.0f : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4_t bq = *reinterpret_cast<const float4_t*>(pb + q * 8);
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = acc[i][j][q * 4 + e] + bq[e] * keep;
            o[e] = (half_t)(p.act ? y5_silu(t) : t);
          }
          *reinterpret_cast<half4_t*>(scratch + frow * SCR_ROWB + (q * 8 + g * 4) * 2) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int rc = o_rc[i][ps];
          const int oh = oh0 + (rc & 0xffff), ow = ow0 + (rc >> 16);
          const int vs = lane & 3;
          const int n = nt + vs * 8;
          if (rc >= 0 && oh < p.OH && ow < p.OW && n < p.C2) {
            uint4_t raw = *reinterpret_cast<const uint4_t*>(scratch + (ps * 16 + (lane >> 2)) * SCR_ROWB + vs * 16);
            const size_t mo = ((size_t)b * p.OH + oh) * p.OW + ow;
            if (rg) {
              const uint4_t rr = *reinterpret_cast<const uint4_t*>(rg + mo * p.ldr + n);
              half8_t a = __builtin_bit_cast(half8_t, raw), r8 = __builtin_bit_cast(half8_t, rr), c;
#pragma unroll
              for (int e = 0; e < 8; ++e) c[e] = (half_t)((float)a[e] + (float)r8[e]);
              raw = __builtin_bit_cast(uint4_t, c);
            }
            *reinterpret_cast<uint4_t*>(yg + mo * p.ldy + n) = raw;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  };

  if (nmine > 0) prologue(0);
  for (int ti = 0; ti < nmine; ++ti) {
    int tb, toh0, tow0, tn0;
    tile_coords(ti, tb, toh0, tow0, tn0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- software-pipelined main loop ---------------------------------------------------------------------------------------------
    // A step multiplies fragments that were read from LDS during the PREVIOUS step: between the two k-step halves of step s the
    // fragments of step s+1 are fetched into the registers the half just consumed, so that the matrix cores never wait for an LDS
    // round trip (measured on the loop skeleton, scripts/ubench/step_skeleton.hip: 1488 -> 998 cycles per 20-MFMA step).  The wait /
    // barrier at the top of step s therefore covers what step s+1 reads: filter slice s+1 (issued seven steps ago, eight ahead of
    // its use) and -- at the last tap of a chunk -- the next chunk's halo (issued in taps 0-1).
    half8_t af[2][TM], wf[2][TN];
    auto frag_addr = [&](int tap, int cc_, int (&a0)[TM]) {
      int hoff = (tap / 3) * HW + (tap % 3) + (cc_ & 1) * (A_STAGE / 64);  // stage base in halo-pixel units (a multiple of 16: swizzle unchanged)
#ifndef Y5_EMU
      asm volatile("" : "+s"(hoff));  // keeps the per-tap fragment addresses out of the loop-invariant set (45+ VGPRs if hoisted)
#endif
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int hp = hp0[i] + hoff;
        a0[i] = (hp << 6) | ((g ^ ((hp >> 2) & 3)) << 4);
      }
    };
    auto read_frags = [&](auto ksc, const int (&a0)[TM], const char* wst) {
      constexpr int ks = decltype(ksc)::value;
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const half8_t*>(wst + (w_rd[j] ^ (ks * 32)));
#pragma unroll
      for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const half8_t*>(a_lds + (a0[i] ^ (ks * 32)));
    };
    {
      // halo chunk 0 and filter slice 0 have landed (the prologue's slices 1..7 may still be in flight); fragments of step 0
      y5_wait_vm<(NSW - 2) * WPW>();
      __builtin_amdgcn_s_barrier();
      int a0[TM];
      frag_addr(0, 0, a0);
      read_frags(std::integral_constant<int, 0>{}, a0, w_lds);
      read_frags(std::integral_constant<int, 1>{}, a0, w_lds);
    }
    for (int cc = 0; cc < NCC; ++cc) {
      const bool last = cc + 1 == NCC;
      y5_static_for<0, 9>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int tn = (t + 1) % 9;
#ifdef Y5_H3_TIMING
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // everything older than the last WIN = NSW-3 steps has landed: filter slice s+1 (issued NSW-2 steps ago or by the prologue) and,
        // at t == 8, halo chunk cc+1 (issued in taps 0-1).  Per step a wave issues WPW filter pieces (while a slice NSW-1 steps ahead
        // exists: always before the last chunk, taps 0..9-NSW inside it) and, before the last chunk, the next halo's pieces in taps 0-1.
        if (last) y5_wait_vm<WPW * Gm::last_w(t)>();
        else y5_wait_vm<Gm::WIN * WPW + Gm::cur_a(t)>();
        __builtin_amdgcn_s_barrier();
#ifdef Y5_H3_TIMING
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (r_first == 0) r_first = __builtin_amdgcn_s_memrealtime();
#endif
        // slice s + NSW - 1 goes into the stage slice s - 1 occupied (every wave finished reading it before this barrier)
        constexpr int t2 = (t + NSW - 1) % 9;
        const int st2 = NSW == 9 ? t2 : (cc + t + NSW - 1) & 3;  // (9 cc + t + NSW - 1) mod NSW
        if (!last) {
          if constexpr (t < 2) y5_static_for<t * Gm::PPS, (t + 1) * Gm::PPS < APS ? (t + 1) * Gm::PPS : APS>([&](auto kc) { issue_a(kc, cc + 1, true); });
          issue_w(st2, t2, t + NSW - 1 >= 9 ? cc + 1 : cc);
        } else if constexpr (t <= 9 - NSW) {
          issue_w(st2, t2, cc);
        }
        const bool more = t < 8 || !last;  // a step s+1 exists in this tile
        int a0[TM];
        if (more) frag_addr(tn, t == 8 ? cc + 1 : cc, a0);
        const char* wst = w_lds + (NSW == 9 ? tn : (cc + t + 1) & 3) * W_STAGE;  // stage of step s + 1
#ifdef Y5_H3_TIMING
        const unsigned long long t2_ = __builtin_amdgcn_s_memtime();
#endif
        y5_static_for<0, 2>([&](auto ksc) {
          constexpr int ks = decltype(ksc)::value;
#ifndef Y5_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
#ifndef Y5_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
          if (more) read_frags(ksc, a0, wst);
        });
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef Y5_H3_TIMING
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        d_wait += t1 - t0; d_issue += t2_ - t1; d_comp += t3 - t2_; ++d_steps;
#endif
      });
    }
#ifdef Y5_H3_TIMING
    r_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();  // every wave is done with the stages: the next tile's first loads may overwrite them while the epilogue runs
    // the next tile's first loads fly during the epilogue -- unless it reads a residual: its first use would wait for every older
    // vector-memory operation (in-order vmcnt), i.e. for the whole prologue
    if (!rg && ti + 1 < nmine) prologue(ti + 1);
    epilogue(tb, toh0, tow0, tn0);
    if (rg && ti + 1 < nmine) prologue(ti + 1);
  }
#ifdef Y5_H3_TIMING
  if (lane == 0 && blockIdx.x == 0) {
    unsigned long long* o = y5_h3_dbg + wave * 8;
    o[0] = d_wait; o[1] = d_issue; o[2] = d_comp; o[3] = d_steps;
  }
  if (tid == 0 && blockIdx.x < 1024) {
    unsigned long long* o = y5_h3_blocks + blockIdx.x * 4;
    o[0] = r_entry; o[1] = r_first; o[2] = r_loop_end; o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}
