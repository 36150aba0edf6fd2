// Fused Bottleneck at c_ = 128 (models/common.py:164-181 inside C3, :230-246; yolov5s: 6.C3.m0-2, 13.C3.m0, 20.C3.m0 at 40 x 40):
//     y = [x +] SiLU(W2 (*) SiLU(W1 x + b1) + b2),   cv1 = 1x1 128 -> 128, cv2 = 3x3 pad 1 128 -> 128 (BN folded)
// as ONE launch built on the halo-resident 3x3 kernel (conv_h3.h).  What the two-launch form pays for -- the 1x1's output t written to HBM
// and read back 1.3x (halo), one kernel boundary -- becomes a GEMM-1 phase in front of the 3x3 loop:
//   * a workgroup owns a TH x TW spatial tile of one image; the (TH+2) x (TW+2) halo of ALL 128 channels lives in LDS as four 32-channel
//     planes (HPMAX = 320 halo pixels x 64 B = 20 KB each; conv_h3.h layout: 16-byte slots XOR-swizzled on the source side);
//   * phase 1: the planes hold the x halo (LDS-DMA straight from the NHWC tensor, out-of-image pixels zero-filled); t = W1 x for every halo pixel
//     (<= 10 row blocks x 4 channel blocks of 32 x 32, K = 128, dealt round-robin: 4-5 accumulator blocks per wave); after a barrier SiLU(t + b1) is written OVER the x halo as fp16 -- exactly the bytes the unfused 1x1 would
//     have stored -- with out-of-image halo pixels forced to 0 (the 3x3's zero padding applies to t, not to x);
//   * phase 2: conv_h3.h's software-pipelined tap loop over the resident planes -- only the 3x3 filter streams (NSW-stage ring of 8 KB slices,
//     counted vmcnt, one barrier per (tap, chunk) step);
//   * the NEXT tile's x halo streams into a plane as soon as the loop has finished with it (plane cc-1 during chunk cc, the last plane behind the
//     loop), so phase 1 of the next tile never waits for HBM; the residual is re-read from x (L2-hot: this workgroup fetched it one tile ago) in the
//     epilogue, whose transposition scratch aliases the idle filter ring.
// LDS: 80 KB planes + 9 x 8 KB ring (W1 streamed through four of its stages between tiles) + 1 KB dummy = 153 KB; with a resident W1 (32 KB) the
// ring has 4-5 stages (145 / 153 KB).  One workgroup of eight waves per CU.
// Halo recompute of the 1x1: (TH+2)(TW+2) / (TH TW) = 1.32x of a GEMM that is 1/9 of the 3x3's work.
#pragma once
#include "conv_h3.h"

struct Y5H3bParams {
  const void* x;            // (B, H, W, >= 128 channels) NHWC slice, pixel stride ldx
  const void* w1;           // [128][Kpad1] fp16, k = c
  const void* w2;           // [128][Kpad2] fp16, k = (kh, kw, c)
  const float* b1; const float* b2;
  void* y;                  // NHWC slice, pixel stride ldy; must not overlap x
  unsigned x_bytes, w1_bytes, w2_bytes;
  int B, H, W, ldx, ldy, Kpad1, Kpad2, add;
  int th, tw, tiles_h, tiles_w;
  // CV3 kernels: C3's cv3 in the same launch (models/common.py:246: cv3(cat(m(cv1(x)), cv2(x)))): out = act3(W3 [y ; y2] + b3); y (the Bottleneck's
  // result) never leaves LDS, `y` above is then the destination `out` (pixel stride ldy, C3 <= 256 channels)
  const void* y2;           // C3.cv2's output: NHWC slice of 128 channels, pixel stride ld2
  const void* w3;           // [256][Kpad3] fp16, k = (y's 128 channels, then y2's)
  const float* b3;
  unsigned y2_bytes, w3_bytes;
  int ld2, Kpad3, C3, act3;
};

template <int NSW_, bool K64_ = false, bool CV3_ = false>
struct Y5H3bGeom {
  static constexpr bool CV3 = CV3_;   // + C3's cv3 as a GEMM-3 phase behind the tap loop (the planes hold the Bottleneck's result as its A operand)
  static constexpr bool K64 = K64_;   // one barrier per TWO (tap, chunk) slices = 16 MFMAs per wave (needs the deep ring: two double-steps of lead)
  static constexpr int C = 128, NCC = 4, NW = 8, WN = 2, TM = 2, TN = 2, HPMAX = 320;
  static constexpr int NAI_MAX = HPMAX / 16, PLANE = NAI_MAX * 1024;   // 20 pieces of 16 halo pixels x 64 B
  static constexpr int W_STAGE = C * 64, NSW = NSW_, WIN = NSW - 3;
  static constexpr int APS = (NAI_MAX + NW - 1) / NW, PPS = (APS + 1) / 2;   // halo pieces per wave per plane; issued in tap 0 / tap 1
  static constexpr int NSTEP = NCC * 9;
  static constexpr int NRB1 = 5;                                        // GEMM-1: wave w owns channel block w & 3 of the halo row blocks (w >> 2) + 2 i
  static constexpr int SCR_ROWB = 32 * 2 + 16, SCR_BYTES = 32 * SCR_ROWB;
  // W1 (32 KB): resident behind a short ring (NSW <= 5), or -- so that the ring can be as deep as conv_h3.h's (NSW = 9: slices issued eight steps
  // ahead; with two steps of lead a one-workgroup-per-CU loop waits on L2 latency, profiles/r05/r05_ab_h3b_v1.log) -- STREAMED per tile into ring
  // stages W1_ST0..W1_ST0+3, which are idle between the last tap of a tile and the end of the next tile's GEMM 1; the epilogue's scratch keeps
  // stages 0..W1_ST0-1
  static constexpr bool W1RES = NSW <= 5;
  static constexpr int NPRO = K64 ? NSW - 2 : NSW - 1;                  // ring slices issued before the tap loop
  static constexpr int W1_ST0 = 3, PRE0 = W1RES ? NPRO : W1_ST0;        // ... of which in front of GEMM 1 (the rest behind it)
  static_assert(!K64 || (NSW == 9 && !W1RES), "double steps: 9-stage ring");
  static constexpr size_t OFF_RING = (size_t)NCC * PLANE;
  static constexpr size_t OFF_W1 = W1RES ? OFF_RING + (size_t)NSW * W_STAGE : OFF_RING + (size_t)W1_ST0 * W_STAGE;
  static constexpr size_t OFF_DUMMY = OFF_RING + (size_t)NSW * W_STAGE + (W1RES ? (size_t)C * C * 2 : 0);
  static constexpr size_t OFF_BIAS = OFF_DUMMY + 1024;   // b1, b2 (, b3) as fp32: per-lane ds_read_b128 instead of 16 global loads per accumulator block
  static constexpr size_t LDS = OFF_BIAS + (CV3 ? 4 : 2) * C * 4;
  // GEMM 3 (CV3): K = 256 in eight 32-channel chunks; chunks 0..3 take their A operand from the planes (the Bottleneck's result), chunks 4..7 stream y2
  // (256 rows x 64 B) beside the filter chunk (256 rows x 64 B): two 32 KB stages inside the ring, B = [40 KB, 72 KB) (filled while the first epilogue still
  // uses the scratch at [0, 20 KB)) and A = [8 KB, 40 KB)
  static constexpr int G3_STAGE = 2 * 256 * 64, G3_OFF_B = 5 * W_STAGE, G3_OFF_A = W_STAGE, G3_PPW = 4;
  static_assert(!CV3 || (NSW == 9 && !K64_), "cv3 phase: 9-stage ring, single-step loop");
  static_assert(NW * SCR_BYTES <= (W1RES ? NSW : W1_ST0) * W_STAGE, "epilogue scratch must fit into the filter ring (in front of the streamed W1)");
  static_assert(NSW >= 4 && NSW <= 9 && (W1RES || NSW >= W1_ST0 + 5) && LDS <= 160 * 1024, "ring depth");
  // pieces of the next tile's halo a wave issues in tap t of any chunk
  static constexpr int xi(int t) { return t == 0 ? PPS : t == 1 ? APS - PPS : 0; }
  // LDS-DMA instructions a wave may still have in flight at the barrier of (chunk, tap t): everything issued in the WIN steps before it.
  // Filter slices: one per step while a slice NSW-1 steps ahead exists (always before the last chunk; taps <= 9 - NSW inside it; the ring prologue
  // counts as the NSW-1 steps in front of the loop); halo pieces: taps 0 and 1 of the current chunk.
  static constexpr int allowed(int t, bool last) {
    int n = 0;
    for (int u = t - WIN; u <= t - 1; ++u) {
      n += (!last || u <= 9 - NSW) ? 1 : 0;
      if (u >= 0) n += xi(u);
    }
    return n;
  }
  // ---- K64: double-step k = 0..8 of a chunk PAIR (18 slices); double-step d = 9 P + k multiplies slices 2d, 2d+1, issues slices 2d+7, 2d+8 (while
  // they exist: d <= 13 both, d = 14 one) and, at k = 0, 1 (plane 1, second pair only) and k = 5, 6 (plane 2 P), the next tile's halo pieces.
  static constexpr int xi2(int k) { return (k == 0 || k == 5) ? PPS : (k == 1 || k == 6) ? APS - PPS : 0; }
  static constexpr int wi2(int k, bool last) { return !last || k <= 4 ? 2 : k == 5 ? 1 : 0; }   // (last: d = 9 + k)
  // in flight at the barrier of double-step k: what the two double-steps before it issued (the ring prologue counts as those in front of the loop)
  static constexpr int allowed2(int k, bool last) {
    int n = 0;
    for (int u = k - 2; u <= k - 1; ++u) {
      n += u < 0 ? 2 : wi2(u, last);   // u < 0: the previous pair (never the last) or the prologue (slices 3..6)
      if (u >= 0) n += xi2(u);
    }
    return n;
  }
};

#ifdef Y5_H3B_TIMING   // kernel-experiment builds only (scripts/h3b_timing.py): s_memrealtime (100 MHz) at the phase boundaries of the first four tiles
__device__ unsigned long long y5_h3b_stamps[512 * 4 * 8];
#define Y5_H3B_STAMP(k) do { if (tid == 0 && blockIdx.x < 512 && ti < 4) y5_h3b_stamps[(blockIdx.x * 4 + ti) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define Y5_H3B_STAMP(k) ((void)0)
#endif

template <int NSW_, bool K64_ = false, bool CV3_ = false>
__global__ __launch_bounds__(512, 1)
void y5_conv_h3b_kernel(const Y5H3bParams p) {
  typedef half_t T;
  using Gm = Y5H3bGeom<NSW_, K64_, CV3_>;
  constexpr bool CV3 = CV3_;
  constexpr int NW = Gm::NW, TM = Gm::TM, TN = Gm::TN, NSW = Gm::NSW, APS = Gm::APS, PLANE = Gm::PLANE, W_STAGE = Gm::W_STAGE;
  constexpr int SCR_ROWB = Gm::SCR_ROWB, NRB1 = Gm::NRB1;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const a_lds = smem;                    // four planes: x halo, then t halo
  char* const w_lds = smem + Gm::OFF_RING;     // 3x3 filter ring; the epilogue's transposition scratch between two tiles
  char* const w1_lds = smem + Gm::OFF_W1;      // 1x1 filter, four 32-channel slices of [128 rows][64 B]
  char* const dummy = smem + Gm::OFF_DUMMY;
  float* const bias_lds = reinterpret_cast<float*>(smem + Gm::OFF_BIAS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 5, frow = lane & 31;
  char* const scratch = w_lds + wave * Gm::SCR_BYTES;

  const int TH = p.th, TW = p.tw, HW = TW + 2;
  const int HP = (TH + 2) * HW;
  const int NAI = (HP + 15) >> 4;
  const int NRB = (HP + 31) >> 5;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t w1rs = y5_make_rsrc(p.w1, p.w1_bytes);
  const y5_rsrc_t w2rs = y5_make_rsrc(p.w2, p.w2_bytes);

  // ---- per-lane constants ------------------------------------------------------------------------------------------------------
  // (a) halo pieces this wave stages per plane: instruction I = k * NW + wave covers 16 halo pixels x 4 slots
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
  // (b) 3x3 fragment reads: lane (pixel row frow of block i, k-half g) -> halo pixel of tap (0, 0); filter row of block j
  int hp0[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = (wm * TM + i) * 32 + frow;
    const int r = m / TW, c = m - r * TW;
    hp0[i] = r < TH ? r * HW + c : 0;
  }
  const int fsw = (frow >> 2) & 3;
  int w_rd[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) w_rd[j] = ((wn * TN + j) * 32 + frow) * 64 + ((g ^ fsw) << 4);
  // (c) epilogue store rows: lane -> tile pixel (pass ps, row ps*16 + lane/4), 16-byte vector lane%4 of the 32 channels
  int o_rc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = (wm * TM + i) * 32 + ps * 16 + (lane >> 2);
      const int r = m / TW, c = m - r * TW;
      o_rc[i][ps] = r < TH ? (r | (c << 16)) : -1;
    }
  // (d) GEMM 1: the <= 40 blocks of 32 halo pixels x 32 channels are dealt round-robin (block b = w + 8 i -> channel block b & 3 = w & 3, row block
  // b >> 2 = (w >> 2) + 2 i): 4-5 blocks per wave at 9 row blocks instead of 4-6 with a 2 x 3 grid; lane -> halo pixel (row, column) for the
  // in-image test of the t store
  const int cb1 = wave & 3, rb1 = wave >> 2;
  const int w1_rd = (cb1 * 32 + frow) * 64 + ((g ^ ((frow >> 2) & 3)) << 4);
  int g1_rc[NRB1];
#pragma unroll
  for (int i = 0; i < NRB1; ++i) {
    const int hp = (rb1 + 2 * i) * 32 + frow;
    const int hr = hp / HW, hc = hp - hr * HW;
    g1_rc[i] = hr | (hc << 8) | ((hp < HP ? 1 : 0) << 16);
  }
  // (e) filter pieces: one 16-row piece per wave per slice (3x3 ring) / per 32-channel slice of W1
  unsigned w2_off, w1_off;
  {
    const int row = wave * 16 + (lane >> 2);
    const int ss = (lane & 3) ^ ((row >> 2) & 3);
    w2_off = (unsigned)(row * p.Kpad2 * 2 + ss * 16);
    w1_off = (unsigned)(row * p.Kpad1 * 2 + ss * 16);
  }

  // ---- tile schedule -----------------------------------------------------------------------------------------------------------
  const int G = gridDim.x, bid = blockIdx.x;
  const int ntiles = p.B * p.tiles_h * p.tiles_w;
  const int nmine = (ntiles - bid + G - 1) / G;
  if (nmine <= 0) return;
  auto tile_coords = [&](int j, int& b, int& oh0, int& ow0) {
    const int t = y5_xcd_remap(bid + j * G, ntiles);
    const int tx = t % p.tiles_w, q = t / p.tiles_w;
    const int ty = q % p.tiles_h;
    b = q / p.tiles_h;
    oh0 = ty * TH;
    ow0 = tx * TW;
  };

  // ---- loader state of the tile whose x halo is being staged -----------------------------------------------------------------------
  int s_base = 0, s_ih0 = 0, s_iw0 = 0;
  bool s_valid = false;
  auto x_setup = [&](int j) {
    s_valid = j < nmine;
    if (s_valid) {
      int b, oh0, ow0;
      tile_coords(j, b, oh0, ow0);
      s_ih0 = oh0 - 1;
      s_iw0 = ow0 - 1;
      s_base = ((b * p.H + s_ih0) * p.W + s_iw0) * p.ldx * 2;
    }
  };
  auto issue_x = [&](auto kc, int cc, bool real) {  // piece k of plane cc of the staged tile; a wave without that piece issues a dummy (uniform counts)
    constexpr int k = decltype(kc)::value;
    const int I = k * NW + wave;
    if (real && s_valid && I < NAI) {
      const int ih = s_ih0 + (a_rc[k] & 0xff), iw = s_iw0 + ((a_rc[k] >> 8) & 0xff);
      const bool ok = (a_rc[k] >> 16) != 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      y5_bglds16(xrs, ok ? (unsigned)(s_base + a_rel[k] + cc * 64) : Y5_OOB, a_lds + cc * PLANE + I * 1024);
    } else {
      y5_bglds16_dummy(xrs, dummy);
    }
  };
  auto issue_w2 = [&](int stage, int s) {  // slice s = chunk * 9 + tap
    const int cc = s / 9, tap = s - cc * 9;
    y5_bglds16(w2rs, w2_off + (unsigned)((tap * Gm::C + cc * 32) * 2), w_lds + stage * W_STAGE + wave * 1024);
  };

  float16_t acc[TM][TN];
  T* yg = static_cast<T*>(p.y);
  const T* rg = p.add ? static_cast<const T*>(p.x) : nullptr;

  // ---- epilogue: bias + SiLU -> per-wave LDS transpose -> (+ x) -> 16-byte row-contiguous stores ---------------------------------------
  auto epilogue = [&](int b, int oh0, int ow0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nt = (wn * TN + j) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4_t bq = *reinterpret_cast<const float4_t*>(bias_lds + Gm::C + nt + q * 8 + g * 4);
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)y5_silu(acc[i][j][q * 4 + e] + bq[e]);
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
          if (rc >= 0 && oh < p.H && ow < p.W) {
            uint4_t raw = *reinterpret_cast<const uint4_t*>(scratch + (ps * 16 + (lane >> 2)) * SCR_ROWB + vs * 16);
            const size_t mo = ((size_t)b * p.H + oh) * p.W + ow;
            if (rg) {
              const uint4_t rr = *reinterpret_cast<const uint4_t*>(rg + mo * p.ldx + n);
              half8_t a = __builtin_bit_cast(half8_t, raw), r8 = __builtin_bit_cast(half8_t, rr), c;
#pragma unroll
              for (int e = 0; e < 8; ++e) c[e] = (half_t)((float)a[e] + (float)r8[e]);
              raw = __builtin_bit_cast(uint4_t, c);
            }
            if constexpr (!CV3) *reinterpret_cast<uint4_t*>(yg + mo * p.ldy + n) = raw;
            else {   // the Bottleneck's result as the A operand of GEMM 3: plane n / 32, row = tile pixel m, 16-byte slot swizzled like every operand tile
              const int m = (wm * TM + i) * 32 + ps * 16 + (lane >> 2);
              *reinterpret_cast<uint4_t*>(a_lds + (n >> 5) * PLANE + (m << 6) + ((vs ^ ((m >> 2) & 3)) << 4)) = raw;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  };

  // ---- launch prologue: W1 (once per workgroup) and the first tile's x halo ------------------------------------------------------------------
  auto issue_w1 = [&]() {
#pragma unroll
    for (int cc = 0; cc < Gm::NCC; ++cc) y5_bglds16(w1rs, w1_off + (unsigned)(cc * 64), w1_lds + cc * W_STAGE + wave * 1024);
  };
  if (tid < Gm::C) {   // (visible to every wave behind the first tile's barrier)
    bias_lds[tid] = p.b1[tid];
    bias_lds[Gm::C + tid] = p.b2[tid];
    if constexpr (CV3) {
      bias_lds[2 * Gm::C + tid] = tid < p.C3 ? p.b3[tid] : 0.f;
      bias_lds[3 * Gm::C + tid] = Gm::C + tid < p.C3 ? p.b3[Gm::C + tid] : 0.f;
    }
  }
  issue_w1();
  x_setup(0);
#pragma unroll
  for (int cc = 0; cc < Gm::NCC; ++cc) y5_static_for<0, APS>([&](auto kc) { issue_x(kc, cc, true); });

  for (int ti = 0; ti < nmine; ++ti) {
    int tb, toh0, tow0;
    tile_coords(ti, tb, toh0, tow0);

    // ---- phase 1: t = SiLU(W1 x + b1) over the halo ---------------------------------------------------------------------------------------
    Y5_H3B_STAMP(0);
    y5_wait_vm<0>();      // this tile's x halo (and W1) landed
    __syncthreads();      // ... for every wave; the previous tile's epilogue is done with the ring
    Y5_H3B_STAMP(1);
    y5_static_for<0, Gm::PRE0>([&](auto uc) { issue_w2(decltype(uc)::value, decltype(uc)::value); });  // the first 3x3 slices fly behind GEMM 1
    {
      float16_t acc1[NRB1];
#pragma unroll
      for (int i = 0; i < NRB1; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
      int a1[NRB1];
#pragma unroll
      for (int i = 0; i < NRB1; ++i) {
        const int hp = (rb1 + 2 * i) * 32 + frow;
        a1[i] = (hp << 6) | ((g ^ ((hp >> 2) & 3)) << 4);
      }
      // all five row blocks are multiplied unconditionally (a block beyond the halo reads whatever the plane holds there and is never stored: the
      // barrier below waits for the waves that own five real blocks anyway); fragments of k-step s+1 are fetched while step s multiplies
      half8_t wf1[2], af1[2][NRB1];
      auto rd1 = [&](auto sc) {
        constexpr int s1 = decltype(sc)::value, cc = s1 >> 1, ks = s1 & 1, bf = s1 & 1;
        wf1[bf] = *reinterpret_cast<const half8_t*>(w1_lds + cc * W_STAGE + (w1_rd ^ (ks * 32)));
#pragma unroll
        for (int i = 0; i < NRB1; ++i) af1[bf][i] = *reinterpret_cast<const half8_t*>(a_lds + cc * PLANE + (a1[i] ^ (ks * 32)));
      };
      rd1(std::integral_constant<int, 0>{});
      y5_static_for<0, Gm::NCC * 2>([&](auto sc) {
        constexpr int s1 = decltype(sc)::value, bf = s1 & 1;
        if constexpr (s1 + 1 < Gm::NCC * 2) rd1(std::integral_constant<int, s1 + 1>{});
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < NRB1; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf1[bf], af1[bf][i], acc1[i], 0, 0, 0);
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
      });
      __syncthreads();    // every wave has read its x fragments: the planes may now take t (and the ring stages of a streamed W1 their slices)
      Y5_H3B_STAMP(2);
      y5_static_for<Gm::PRE0, Gm::NPRO>([&](auto uc) { issue_w2(decltype(uc)::value, decltype(uc)::value); });
#pragma unroll
      for (int i = 0; i < NRB1; ++i) {
        if (rb1 + 2 * i < NRB) {
          const int hp = (rb1 + 2 * i) * 32 + frow;
          const int ih = toh0 - 1 + (g1_rc[i] & 0xff), iw = tow0 - 1 + ((g1_rc[i] >> 8) & 0xff);
          const bool ok = (g1_rc[i] >> 16) != 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          const int sw = (hp >> 2) & 3;
          const uint32_t keep = ok ? 0xffffffffu : 0u;   // branch-free: the zero padding of the 3x3 is a bit mask on the packed pair
          char* row = a_lds + cb1 * PLANE + (hp << 6) + g * 8;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4_t bq = *reinterpret_cast<const float4_t*>(bias_lds + cb1 * 32 + q * 8 + g * 4);
            uint2_t o;
            o[0] = y5_pack_h2(y5_silu(acc1[i][q * 4 + 0] + bq[0]), y5_silu(acc1[i][q * 4 + 1] + bq[1])) & keep;
            o[1] = y5_pack_h2(y5_silu(acc1[i][q * 4 + 2] + bq[2]), y5_silu(acc1[i][q * 4 + 3] + bq[3])) & keep;
            *reinterpret_cast<uint2_t*>(row + ((q ^ sw) << 4)) = o;
          }
        }
      }
    }
    x_setup(ti + 1);   // from here on the loader stages the NEXT tile's x halo

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- phase 2: the 3x3 over the resident t halo (conv_h3.h's software-pipelined tap loop) ---------------------------------------------
    half8_t af[2][TM], wf[2][TN];
    auto frag_addr = [&](int tap, int cc_, int (&a0)[TM]) {
      int hoff = (tap / 3) * HW + (tap % 3) + cc_ * (PLANE / 64);  // plane base in halo-pixel units (a multiple of 16: swizzle unchanged)
#ifndef Y5_EMU
      asm volatile("" : "+s"(hoff));
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
      y5_wait_vm<Gm::NPRO - 1>();   // filter slice 0 landed (the rest of the ring prologue may still be in flight)
      __syncthreads();           // ... and every wave's t rows are visible
      Y5_H3B_STAMP(3);
      int a0[TM];
      frag_addr(0, 0, a0);
      read_frags(std::integral_constant<int, 0>{}, a0, w_lds);
      read_frags(std::integral_constant<int, 1>{}, a0, w_lds);
    }
    int st_cur = 0;   // ring stage of the current (first) slice of the step (slice index mod NSW)
    if constexpr (Gm::K64) {
      auto half_step = [&](const int (&a0)[TM], const char* wst, bool more) {
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
      };
      auto wrap = [](int v) { return v >= NSW ? v - NSW : v; };
      for (int P = 0; P < Gm::NCC / 2; ++P) {
        const bool last = P + 1 == Gm::NCC / 2;
        y5_static_for<0, 9>([&](auto kc_) {
          constexpr int k = decltype(kc_)::value;
          constexpr int uB = 2 * k + 1, uN = 2 * k + 2;           // second slice of this double-step / first slice of the next (pair-relative)
          // slices <= 2 d + 2 have landed: the second slice of this double-step and the first of the next, whose fragments are read below
          if (last) y5_wait_vm<Gm::allowed2(k, true)>();
          else y5_wait_vm<Gm::allowed2(k, false)>();
          __builtin_amdgcn_s_barrier();
          // dead planes take the next tile's x halo: plane 2 P once slice 9 (2 P) + 8 has been read (k >= 4), plane 1 at the top of the second pair
          if constexpr (k == 0 || k == 1)
            y5_static_for<k * Gm::PPS, (k + 1) * Gm::PPS < APS ? (k + 1) * Gm::PPS : APS>([&](auto kc) { issue_x(kc, 1, P > 0); });
          if constexpr (k == 5 || k == 6)
            y5_static_for<(k - 5) * Gm::PPS, (k - 4) * Gm::PPS < APS ? (k - 4) * Gm::PPS : APS>([&](auto kc) { issue_x(kc, 2 * P, true); });
          // slices 2 d + 7, 2 d + 8 go into the stages slices 2 d - 2, 2 d - 1 occupied (every wave finished reading them before this barrier)
          const int s0 = 18 * P + 2 * k + NSW - 2;
          if (!last || k <= 4) { issue_w2(wrap(st_cur + NSW - 2), s0); issue_w2(wrap(st_cur + NSW - 1), s0 + 1); }
          else if (k == 5) issue_w2(wrap(st_cur + NSW - 2), s0);
          int a0[TM];
          frag_addr(uB % 9, 2 * P + uB / 9, a0);
          half_step(a0, w_lds + wrap(st_cur + 1) * W_STAGE, true);
          const bool more = k < 8 || !last;
          if (more) frag_addr(uN % 9, 2 * P + uN / 9, a0);
          half_step(a0, w_lds + wrap(st_cur + 2) * W_STAGE, more);
          st_cur = wrap(st_cur + 2);
        });
      }
    } else {
      for (int cc = 0; cc < Gm::NCC; ++cc) {
        const bool last = cc + 1 == Gm::NCC;
        y5_static_for<0, 9>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int tn = (t + 1) % 9;
          if (last) y5_wait_vm<Gm::allowed(t, true)>();
          else y5_wait_vm<Gm::allowed(t, false)>();
          __builtin_amdgcn_s_barrier();
          // plane cc-1 is dead (its last fragments were consumed in the step before this chunk's tap 0): it takes the next tile's x halo
          if constexpr (t < 2)   // (CV3: the planes are needed for GEMM 3 -- dummies keep the counts; the halo is issued behind GEMM 3)
            y5_static_for<t * Gm::PPS, (t + 1) * Gm::PPS < APS ? (t + 1) * Gm::PPS : APS>([&](auto kc) { issue_x(kc, cc - 1, cc > 0 && !CV3); });
          // slice s + NSW - 1 goes into the stage slice s - 1 occupied (every wave finished reading it before this barrier)
          const int st_prev = st_cur == 0 ? NSW - 1 : st_cur - 1;
          if (!last || t <= 9 - NSW) issue_w2(st_prev, cc * 9 + t + NSW - 1);
          const int st_next = st_cur == NSW - 1 ? 0 : st_cur + 1;
          const bool more = t < 8 || !last;  // a step s+1 exists in this tile
          int a0[TM];
          if (more) frag_addr(tn, t == 8 ? cc + 1 : cc, a0);
          const char* wst = w_lds + st_next * W_STAGE;
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
          st_cur = st_next;
        });
      }
    }
    __syncthreads();  // every wave is done with the ring and with the last plane
    Y5_H3B_STAMP(4);
    if constexpr (!CV3) {
      y5_static_for<0, APS>([&](auto kc) { issue_x(kc, Gm::NCC - 1, true); });  // the next tile's last plane flies during the epilogue
      if constexpr (!Gm::W1RES) {
        if (ti + 1 < nmine) issue_w1();   // ... and so does its W1, into the ring stages behind the scratch
      }
      epilogue(tb, toh0, tow0);
    } else {
      // ---- phase 3: out = act3(W3 [y ; y2] + b3) ---------------------------------------------------------------------------------------------
      const y5_rsrc_t y2rs = y5_make_rsrc(p.y2, p.y2_bytes);
      const y5_rsrc_t w3rs = y5_make_rsrc(p.w3, p.w3_bytes);
      // loader: per chunk a wave issues two 16-row pieces of the filter chunk and (chunks 4..7) two of y2's rows = tile pixels
      unsigned w3_off[2], y2_off[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = (q * NW + wave) * 16 + (lane >> 2);
        const int ss = (lane & 3) ^ ((row >> 2) & 3);
        w3_off[q] = row < p.C3 ? (unsigned)(row * p.Kpad3 * 2 + ss * 16) : Y5_OOB;
        const int r = row / TW, c = row - r * TW;
        const bool ok = r < TH && toh0 + r < p.H && tow0 + c < p.W;
        y2_off[q] = ok ? (unsigned)((((tb * p.H + toh0 + r) * p.W + tow0 + c) * p.ld2) * 2 + ss * 16) : Y5_OOB;
      }
      // (no dummies here: chunks 0..3 are two loads per wave, chunks 4..7 four, and the waits below count exactly that.  The first version padded
      // with pairs of plain out-of-range loads into the dummy region, which the compiler merged into ONE load -- every counted wait behind them one short, wrong results at
      // slow y2 pitches: profiles/r05/r05_dummy_dma_merge.log, y5_common.h y5_bglds16_dummy)
      auto issue3 = [&](int c) {
        char* st = w_lds + ((c & 1) ? Gm::G3_OFF_A : Gm::G3_OFF_B);
#pragma unroll
        for (int q = 0; q < 2; ++q) y5_bglds16(w3rs, w3_off[q] == Y5_OOB ? Y5_OOB : w3_off[q] + (unsigned)(c * 64), st + (q * NW + wave) * 1024);
        if (c >= 4) {
#pragma unroll
          for (int q = 0; q < 2; ++q) y5_bglds16(y2rs, y2_off[q] == Y5_OOB ? Y5_OOB : y2_off[q] + (unsigned)((c - 4) * 64), st + 256 * 64 + (q * NW + wave) * 1024);
        }
      };
      issue3(0);                       // into stage B: flies during the first epilogue (whose scratch is below it)
      epilogue(tb, toh0, tow0);        // y = [x +] SiLU(acc + b2) -> planes (fp16, exactly what the two-launch form stores)
      __syncthreads();
      float16_t acc3[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.f;
      int a3[2], w3_rd[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = (wm * 2 + i) * 32 + frow;
        a3[i] = (m << 6) | ((g ^ ((m >> 2) & 3)) << 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) w3_rd[j] = ((wn * 4 + j) * 32 + frow) * 64 + ((g ^ fsw) << 4);
      for (int c = 0; c < 8; ++c) {
        if (c + 1 < 8) issue3(c + 1);
        // chunk c landed (chunk c + 1 may be in flight: two loads per wave up to chunk 3, four from chunk 4 on)
        if (c + 1 < 4) y5_wait_vm<2>();
        else if (c + 1 < 8) y5_wait_vm<4>();
        else y5_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const char* st = w_lds + ((c & 1) ? Gm::G3_OFF_A : Gm::G3_OFF_B);
        const char* ab = c < 4 ? a_lds + c * PLANE : st + 256 * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          half8_t af3[2], wf3[4];
#pragma unroll
          for (int i = 0; i < 2; ++i) af3[i] = *reinterpret_cast<const half8_t*>(ab + (a3[i] ^ (ks * 32)));
#pragma unroll
          for (int j = 0; j < 4; ++j) wf3[j] = *reinterpret_cast<const half8_t*>(st + (w3_rd[j] ^ (ks * 32)));
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf3[j], af3[i], acc3[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();   // every wave is done with this stage before chunk c + 2 lands in it
      }
      // planes and ring are idle: the next tile's x halo and W1 fly during the second epilogue
#pragma unroll
      for (int cc = 0; cc < Gm::NCC; ++cc) y5_static_for<0, APS>([&](auto kc) { issue_x(kc, cc, true); });
      if (ti + 1 < nmine) issue_w1();
      // second epilogue: bias + act3 -> scratch -> 16-byte row-contiguous stores of the 256-channel result
      T* og = static_cast<T*>(p.y);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nt = (wn * 4 + j) * 32;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4_t bq = *reinterpret_cast<const float4_t*>(bias_lds + 2 * Gm::C + nt + q * 8 + g * 4);
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = acc3[i][j][q * 4 + e] + bq[e]; o[e] = (half_t)(p.act3 ? y5_silu(t) : t); }
            *reinterpret_cast<half4_t*>(scratch + frow * SCR_ROWB + (q * 8 + g * 4) * 2) = o;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int rc = o_rc[i][ps];
            const int oh = toh0 + (rc & 0xffff), ow = tow0 + (rc >> 16);
            const int vs = lane & 3;
            const int n = nt + vs * 8;
            if (rc >= 0 && oh < p.H && ow < p.W && n < p.C3) {
              const uint4_t raw = *reinterpret_cast<const uint4_t*>(scratch + (ps * 16 + (lane >> 2)) * SCR_ROWB + vs * 16);
              const size_t mo = ((size_t)tb * p.H + oh) * p.W + ow;
              *reinterpret_cast<uint4_t*>(og + mo * p.ldy + n) = raw;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    Y5_H3B_STAMP(5);
  }
}
