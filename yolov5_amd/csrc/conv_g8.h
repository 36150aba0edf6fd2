// NHWC implicit-GEMM convolution for the MFMA-bound layers on the 256-row / 8-phase "ping-pong" structure (round 6; configuration ids 95 ..).
//
// Replaces `Conv.forward_fuse` (models/common.py:90-92: act(conv(x)), BN folded by utils/torch_utils.py:224-254) for layers with C1 % 64 == 0 and a long K
// (3x3 s1 / s2 and deep 1x1: 5 / 7 / 21.Conv, SPPF.cv2, Bottleneck.cv2 of yolov5s; the 3x3 320 -> 320 / 640 -> 640 layers of yolov5x), the residual add of
// `Bottleneck.forward` (common.py:181) and the concat-free channel-slice / split stores of conv_igemm.h.
//
// Every other convolution kernel of this library shares one loop scheme: 128-row tiles, ONE barrier per K chunk, 4-8 MFMAs per wave between barriers, every
// wave loading and multiplying in lock step.  None of them gets past 0.34 of the MFMA peak.  This kernel is built the other way round:
//   * workgroup tile 256 pixels x 256 channels, K tile 64 (128-byte LDS rows), eight waves as 2 (pixels) x 4 (channels): a wave owns 128 pixels x 64 channels
//     = 4 x 2 accumulator blocks of mfma_f32_32x32x16_f16 (128 accumulator registers), 32 MFMAs per K tile;
//   * a K tile is FOUR half-tiles of 128 rows x 128 B = 16 KiB (two LDS-DMA instructions per wave each): Act-h0 / Act-h1 = pixel rows {0-63, 128-191} /
//     {64-127, 192-255} (the first / second 64 pixels of BOTH wave rows), Wgt-h0 / Wgt-h1 = the first / second 32 channels of all four wave columns;
//     two K tiles are resident (2 x 64 KiB);
//   * a K tile is multiplied in FOUR phases, one 64-pixel x 32-channel quadrant x K = 64 each (8 MFMAs = 256 matrix-pipe cycles):
//         q1: read Wgt-h0 (4 ds_read_b128) then Act-h0 (8)   stage Act-h1 of K tile t+1      MFMA act0 x wgt0
//         q2: read Wgt-h1 (4)                                stage Wgt-h0 of K tile t+2      MFMA act0 x wgt1
//         q3: read Act-h1 (8)                                stage Act-h0 of K tile t+2      MFMA act1 x wgt1
//         q4: --                                             stage Wgt-h1 of K tile t+2      MFMA act1 x wgt0        + s_waitcnt vmcnt(6)
//     every phase is  { ds_reads ; 2 LDS-DMA ; s_barrier ; lgkmcnt(0) ; s_setprio 1 ; 8 MFMAs ; s_setprio 0 ; s_barrier };
//   * the two wave ROWS run staggered by one barrier (wave row 1 takes one extra s_barrier before the loop): on every SIMD one wave multiplies while its
//     partner reads fragments and issues LDS-DMA -- the matrix pipe sees MFMAs back to back, and s_setprio has something to arbitrate;
//   * vmcnt is NEVER 0 in the loop: the one counted wait per K tile (q4, vmcnt(6)) leaves the three youngest half-tiles in flight across every barrier.
//
// Hazard bookkeeping (cdna_hip_programming.md "The 256^2 8-phase template"; intervals between consecutive s_barriers are numbered I0, I1, ...; wave row 0 reads
// in I(2p-2) and multiplies in I(2p-1), wave row 1 reads in I(2p-1) and multiplies in I(2p)):
//   RAW  the q4 wait of K tile t retires everything older than the three half-tiles staged in q2..q4 of t, i.e. ALL of K tile t+1 (staged q2..q4 of t-1 and
//        q1 of t); both wave rows have executed it before the barrier that opens the first read of t+1 (q1 of t+1): "read one phase after the wait".
//   WAR  a half-tile is restaged two phases after the phase that read it (q3 restages what q1 read, q4 what q2 read, q1 what q3 read): the reads of the LATER wave
//        row (issued in I(2p-1), returned by the lgkmcnt(0) inside I(2p)) are complete before the EARLIER row's LDS-DMA of phase p+2 (I(2p+2)).  One exception,
//        as in the template: Wgt-h0 is read first in q1 (four reads, then a scheduling barrier, then the eight activation reads) and retired by lgkmcnt(8)
//        BEFORE q1's first barrier, so q2 may restage it one phase later.
// The schedule runs seamlessly across the output tiles of a persistent workgroup (the K tiles of all its tiles form one sequence; the next tile's first K tiles
// fly during the epilogue); past the last K tile every lane's offsets are out of range (zero fill, no traffic), so the counted wait means the same thing in every
// iteration.  Global stores of the epilogue also count in vmcnt: loads retire in order among themselves, so a younger store can only make the counted wait
// stricter, never too lenient.
//
// Epilogue WITHOUT an LDS transpose: the filter rows of a 32-channel fragment are staged in a permuted order (row i = 8q + 4g + e of the MFMA holds channel
// (q >> 1) * 16 + g * 8 + (q & 1) * 4 + e), so that after the MFMA lane (pixel = lane & 31, g = lane >> 5) owns accumulator register r = channel
// (r >> 3) * 16 + g * 8 + (r & 7): two runs of eight CONSECUTIVE channels = two 16-byte stores per fragment, bias (fp32, staged once in LDS) read as float4.
//
// LDS row swizzle and the im2col loader are conv_igemm.h's: 16-byte slot s of row r is stored at slot s ^ ((r >> 1) & 7), applied on the per-lane GLOBAL
// source (LDS-DMA destinations are lane-linear); padding taps / pixel tails / channel tails carry bit 31 in their buffer offset = zero fill.
#pragma once
#include "conv_igemm.h"

struct Y5G8Geom {
  static constexpr int NW = 8, BM = 256, BN = 256, BK = 64;
  static constexpr int HT = 128 * 128;              // one half-tile: 128 LDS rows of 128 B
  static constexpr int BUF = 4 * HT;                // one K tile: [Act-h0][Act-h1][Wgt-h0][Wgt-h1]
  static constexpr int MAXN = 2048;                 // bias staged for the whole layer
  static constexpr int OFF_BIAS = 2 * BUF;
  static constexpr size_t LDS = OFF_BIAS + MAXN * 4;
  static constexpr int LPH = 2;                     // LDS-DMA instructions per wave per half-tile
  static_assert(LDS <= 160 * 1024, "LDS budget");
  // MFMA row i (0..31) of a filter fragment <-> channel inside its 32-channel group
  __host__ __device__ static constexpr int chan_of_row(int i) { return ((i >> 4) & 1) * 16 + ((i >> 2) & 1) * 8 + ((i >> 3) & 1) * 4 + (i & 3); }
};


// One staged activation row of the im2col loader: output pixel m -> byte offset of its (b, ih0, iw0) origin + the lane's source slot, and the tap mask (bit
// kh * KW + kw SET <=> that tap lies outside the image, or the pixel is past M): conv_igemm.h's scheme (exact multiply-high division, two bit ranges)
template <bool UP2 = false>
__device__ __forceinline__ void y5_g8_act_row(const Y5ConvParams& p, int m, int sslot, int& base, unsigned& mask, int* base2 = nullptr) {
  const int ohw = p.OH * p.OW;
  const int mm = m < p.M ? m : 0;
  const int b = (int)y5_fastdiv((unsigned)mm, p.dv_ohw_m, p.dv_ohw_s);
  const int r = mm - b * ohw;
  const int oh = (int)y5_fastdiv((unsigned)r, p.dv_ow_m, p.dv_ow_s), ow = r - oh * p.OW;
  const int ih0 = oh * p.SH - p.PH, iw0 = ow * p.SW - p.PW;
  base = (((b * p.H + ih0) * p.W + iw0) * p.ldx + sslot * 8) * 2;
  // UP2 (1x1 s1 layers behind `nn.Upsample(2)` + `Concat`, conv_igemm.h): the pixel (b, oh >> 1, ow >> 1) of the low-resolution tensor that supplies the
  // input channels [0, up_c)
  if constexpr (UP2) *base2 = (((b * (p.H >> 1) + (oh >> 1)) * (p.W >> 1) + (ow >> 1)) * p.ldx2 + sslot * 8) * 2;
  unsigned mk = 0;
  if (m < p.M) {
    auto range_bits = [](int lo, int hi) __attribute__((always_inline)) -> unsigned {   // bits [lo, hi), 0 <= lo, hi <= 32
      if (hi <= lo) return 0u;
      const unsigned top = hi >= 32 ? ~0u : (1u << hi) - 1u;
      return top & ~((1u << lo) - 1u);
    };
    const int lo_h = ih0 < 0 ? -ih0 : 0, hi_h = p.H - ih0 < p.KH ? p.H - ih0 : p.KH;
    const int lo_w = iw0 < 0 ? -iw0 : 0, hi_w = p.W - iw0 < p.KW ? p.W - iw0 : p.KW;
    const unsigned bh = range_bits(lo_h < 32 ? lo_h : 32, hi_h), bw = range_bits(lo_w < 32 ? lo_w : 32, hi_w);
    for (int kh = 0; kh < p.KH; ++kh)
      if ((bh >> kh) & 1u) mk |= bw << (kh * p.KW);
  }
  mask = ~mk;
}

#ifndef Y5_G8_ST_AUX
#define Y5_G8_ST_AUX 0   // cache policy of the epilogue stores.  Kernel experiments: 2 = nt measured 1.2-2.9x SLOWER (profiles/r06/r06_stream_probe_v2_taps_nt.log:
                         // the 32-byte pieces of a cache line arrive from four waves and are no longer merged in L2)
#endif

#ifdef Y5_G8_TIMING
// kernel-experiment builds only: per workgroup (< 512) and wave row, stamps of the FIRST output tile -- s_memrealtime (100 MHz) at kernel entry / bias staged /
// K tile 0 landed / K loop done / epilogue done, and the shader-clock length of the K loop (s_memtime)
__device__ unsigned long long y5_g8_dbg[512 * 2 * 8];
#define Y5_G8_STAMP(i) do { if (dbg_on) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define Y5_G8_STAMP(i) ((void)0)
#endif

// SEQ: the K loop walks the taps in the order of Y5ConvParams::tap_seq (stride-2 3x3 launches) instead of the natural (kh, kw) order.  A separate instantiation:
// the look-up walker costs a few scalar instructions per K tile inside the phase that holds the counted wait, and with it compiled into every launch the forward
// lost 2.6 % (profiles/r06/r06_ab_tapseq_runtime_walker.log) -- the other layers keep the incremental walker untouched.
// GEN: input channel counts that are not multiples of the 64-channel K tile (C1 % 8 == 0, C1 >= 64: yolov5x's 80 / 160, yolov5m's 96 / 192).  K stays the
// contiguous (kh, kw, c) axis of the packed filter, so a K tile may begin inside one tap and end inside the next: the first n_lo 16-byte slots of a staged row
// come from tap t at channel c_start + 8 s, the others from tap t + 1 at channel 8 (s - n_lo).  A slot never straddles a tap (C1 % 8 == 0) and a tile never spans
// three (C1 >= 64).  Two scalar (offset, mask bit) pairs per K tile, one per-lane select per load; K tiles past the last tap (Kpad > K) name tap KH * KW, whose
// mask bit is always set (zero fill) -- the filter is zero there as well.  Again a separate instantiation: the C1 % 64 == 0 layers run the code they ran before.
template <bool UP2, bool SEQ = false, bool GEN = false>
__global__ __launch_bounds__(512, 2)
void y5_conv_g8_kernel(const Y5ConvParams p) {
  static_assert(!GEN || (!UP2 && !SEQ), "the general-C1 loader serves plain layers in natural tap order");
  typedef half_t T;
  using Gm = Y5G8Geom;
  constexpr int HT = Gm::HT, BUF = Gm::BUF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const blds = reinterpret_cast<float*>(smem + Gm::OFF_BIAS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 5, l31 = lane & 31;

  const int G = gridDim.x, bid = blockIdx.x;
  const int ntiles = p.tilesM * p.tilesN, nk = p.nk;
  const int nmine = (ntiles - bid + G - 1) / G;   // host guarantees G <= ntiles
#ifdef Y5_G8_TIMING
  const bool dbg_on = (tid & 255) == 0 && bid < 512;
  unsigned long long* const dbg = y5_g8_dbg + (bid < 512 ? bid : 0) * 16 + wr * 8;
  unsigned long long dbg_c0 = 0;
  Y5_G8_STAMP(0);
#endif

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t wrs = y5_make_rsrc(p.w, p.w_bytes);
  const y5_rsrc_t xrs2 = UP2 ? y5_make_rsrc(p.x2, p.x2_bytes) : xrs;

  float bv[Gm::MAXN / 512];   // the bias loads are issued ahead of the prologue's LDS-DMA and written to LDS behind it (older loads retire first)
#pragma unroll
  for (int q = 0; q < Gm::MAXN / 512; ++q) bv[q] = tid + q * 512 < p.Npad ? p.bias[tid + q * 512] : 0.f;
  Y5_G8_STAMP(1);

  // ---- loader: the K tile being staged is (output tile s_t of this workgroup, chunk s_kc); per lane TWO rows (one per LDS-DMA instruction) of each half-tile ----
  const int lrow = lane >> 3, lslot = lane & 7;
  int a_base[2][2];        // [half][instruction]: byte offset of the pixel's (b, ih0, iw0) + source slot
  int a_base2[UP2 ? 2 : 1][2];   // UP2: the same for the low-resolution pixel
  bool st_up = false;
  unsigned a_mask[2][2];   // bit (kh * KW + kw) SET <=> that tap of this pixel lies outside the image (or the pixel is past M)
  unsigned w_off[2][2];    // byte offset of the filter row + source slot; bit 31 for rows past Npad
  int u_kh = 0, u_kw = 0, u_c0 = 0, u_ts = 0, s_t = 0, s_kc = 0;
  int st_tap_off = 0, st_tap_bit = 0;
  unsigned st_kcb = 0;
  int st_off_hi = 0, st_nlo = 8, nx_off_hi = 0, nx_nlo = 8;   // GEN: the K tile's second tap (offset already minus n_lo slots) and its first slot
  const int g_sl0 = lslot ^ ((((wave * 2 + 0) * 8 + lrow) >> 1) & 7), g_sl1 = lslot ^ ((((wave * 2 + 1) * 8 + lrow) >> 1) & 7);   // GEN: this lane's source slot per instruction
  auto gen_hi = [&](int& off_hi, int& nlo) __attribute__((always_inline)) {   // from (u_kh, u_kw, u_c0)
    nlo = (p.C1 - u_c0) >> 3;
    nlo = nlo < 8 ? nlo : 8;
    int kh1 = u_kh, kw1 = u_kw + 1;
    if (kw1 == p.KW) { kw1 = 0; ++kh1; }
    off_hi = ((kh1 * p.W + kw1) * p.ldx) * 2 - nlo * 16;
  };
  const int kw_r = (256 + p.KW - 1) / p.KW;   // t / KW == (t * kw_r) >> 8 for t < 32, KW <= 7 (the tap-sequence walk)
  auto tap_at = [&](int ts) __attribute__((always_inline)) {   // position ts of the launch's tap sequence -> (u_kh, u_kw)
    const int t = (int)((unsigned)(p.tap_seq >> (ts * 4)) & 15u);
    u_kh = (t * kw_r) >> 8;
    u_kw = t - u_kh * p.KW;
  };
#ifdef Y5_G8_ABL_NODMA
  bool t_loop = false;   // ablation build: no LDS-DMA inside the K loop (the prologue's tiles are multiplied over and over)
#endif

  auto tile_coords = [&](int j, int& m0, int& n0) {
    const int t = y5_xcd_remap(bid + j * G, ntiles);
    const int tn = t % p.tilesN, tm = t / p.tilesN;
    m0 = tm * Gm::BM;
    n0 = tn * Gm::BN;
  };
  auto loader_setup = [&](int j) __attribute__((always_inline)) {
    int m0, n0;
    tile_coords(j, m0, n0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int rr = (wave * 2 + jj) * 8 + lrow;             // LDS row inside the half-tile
      const int sslot = lslot ^ ((rr >> 1) & 7);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        y5_g8_act_row<UP2>(p, m0 + (rr >> 6) * 128 + s * 64 + (rr & 63), sslot, a_base[s][jj], a_mask[s][jj], UP2 ? &a_base2[s][jj] : nullptr);
        const int n = n0 + (rr >> 5) * 64 + s * 32 + Gm::chan_of_row(rr & 31);
        w_off[s][jj] = n < p.Npad ? (unsigned)((n * p.Kpad + sslot * 8) * 2) : 0x80000000u;
      }
    }
  };
  auto loader_kill = [&]() {   // past the last K tile: every offset out of range (zero fill, no traffic) -- the counted waits keep their meaning
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) { a_mask[s][jj] = ~0u; w_off[s][jj] = 0x80000000u; }
  };
  auto tap_update = [&]() {
    if constexpr (SEQ) tap_at(0);
    st_tap_off = ((u_kh * p.W + u_kw) * p.ldx + u_c0) * 2;
    st_tap_bit = u_kh * p.KW + u_kw;
    st_kcb = SEQ ? (unsigned)((st_tap_bit * p.C1 + u_c0) * 2) : (unsigned)(s_kc * 128);
    if constexpr (GEN) gen_hi(st_off_hi, st_nlo);
    if constexpr (UP2) st_up = u_c0 < p.up_c;   // (1x1 layer: st_tap_off is the chunk's channel offset; up_c % 64 == 0, a chunk never straddles the boundary)
  };
  // The loader's step to the next K tile in two halves (round 6, after the phase stamps): the SCALAR walker runs in q4, the phase with the shortest load segment
  // (two LDS-DMA instructions and the counted wait), and leaves the next K tile's offsets in nx_*; q1 -- the longest segment: twelve fragment reads -- only
  // commits them behind its own staging (and re-derives the per-lane rows when the sequence enters the next output tile).
  int nx_tap_off = 0, nx_tap_bit = 0;
  unsigned nx_kcb = 0;
  bool nx_up = false, nx_new = false;
  auto advance_scalar = [&]() __attribute__((always_inline)) {
    u_c0 += 64;
    if constexpr (SEQ) {
      nx_new = false;
      if (++s_kc == nk) {
        s_kc = 0; u_c0 = 0; u_ts = 0;
        ++s_t;
        nx_new = true;
        tap_at(0);
      } else if (u_c0 >= p.C1) {
        u_c0 = 0;
        tap_at(++u_ts);
      }
    } else {
      if (u_c0 >= p.C1) {
        u_c0 = GEN ? u_c0 - p.C1 : 0;
        if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
      }
      nx_new = false;
      if (++s_kc == nk) {
        s_kc = 0; u_kh = 0; u_kw = 0; u_c0 = 0;
        ++s_t;
        nx_new = true;
      }
    }
    nx_tap_off = ((u_kh * p.W + u_kw) * p.ldx + u_c0) * 2;
    nx_tap_bit = u_kh * p.KW + u_kw;
    nx_kcb = SEQ ? (unsigned)((nx_tap_bit * p.C1 + u_c0) * 2) : (unsigned)(s_kc * 128);
    if constexpr (GEN) gen_hi(nx_off_hi, nx_nlo);
    if constexpr (UP2) nx_up = u_c0 < p.up_c;
  };
  auto advance_commit = [&]() __attribute__((always_inline)) {
    if (nx_new) {
      if (s_t < nmine) loader_setup(s_t);
      else loader_kill();
    }
    st_tap_off = nx_tap_off; st_tap_bit = nx_tap_bit; st_kcb = nx_kcb;
    if constexpr (GEN) { st_off_hi = nx_off_hi; st_nlo = nx_nlo; }
    if constexpr (UP2) st_up = nx_up;
  };
  auto advance = [&]() __attribute__((always_inline)) { advance_scalar(); advance_commit(); };   // (prologue)
  auto stage_act = [&](auto sc, char* buf) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
#ifdef Y5_G8_ABL_NODMA
    if (t_loop) return;
#endif
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      if constexpr (UP2) {
        if (st_up) {   // wave-uniform
          y5_bglds16(xrs2, (unsigned)(a_base2[s][jj] + st_tap_off) | (a_mask[s][jj] << 31), buf + s * HT + (wave * 2 + jj) * 1024);
          continue;
        }
      }
      unsigned voff;
      if constexpr (GEN) {
        const bool hi = (jj ? g_sl1 : g_sl0) >= st_nlo;
        voff = (unsigned)(a_base[s][jj] + (hi ? st_off_hi : st_tap_off)) | ((a_mask[s][jj] >> (hi ? st_tap_bit + 1 : st_tap_bit)) << 31);
      } else {
        voff = (unsigned)(a_base[s][jj] + st_tap_off) | ((a_mask[s][jj] >> st_tap_bit) << 31);
      }
      y5_bglds16(xrs, voff, buf + s * HT + (wave * 2 + jj) * 1024);
    }
  };
  auto stage_wgt = [&](auto sc, char* buf) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
#ifdef Y5_G8_ABL_NODMA
    if (t_loop) return;
#endif
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) y5_bglds16(wrs, w_off[s][jj] + st_kcb, buf + 2 * HT + s * HT + (wave * 2 + jj) * 1024);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- fragment reads: lane -> row (lane & 31), 16-byte slot (2 ks + g) ^ swz(row); swz depends on lane & 31 only (fragment bases are multiples of 32) ----
  const int fsw = (l31 >> 1) & 7;
  const int a_rd = (wr * 64 + l31) * 128 + ((g ^ fsw) << 4);            // + pf * 4096 + half * HT ; ^ (ks * 32)
  const int w_rd = 2 * HT + (wc * 32 + l31) * 128 + ((g ^ fsw) << 4);   // + half * HT ; ^ (ks * 32)
  half8_t af[2][4], wf[2][4];
  float16_t acc[4][2];   // [half * 2 + pixel fragment][channel half]
#ifdef Y5_G8_ABL_NORD
  bool t_rd = false;     // ablation build: fragments are read in the first K tile only
#endif

  auto rd_act = [&](auto sc, const char* buf) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
#ifdef Y5_G8_ABL_NORD
    if (t_rd) return;
#endif
#pragma unroll
    for (int pf = 0; pf < 2; ++pf)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[pf][ks] = *reinterpret_cast<const half8_t*>(buf + s * HT + pf * 4096 + (a_rd ^ (ks * 32)));
  };
  auto rd_wgt = [&](auto sc, const char* buf) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
#ifdef Y5_G8_ABL_NORD
    if (t_rd) return;
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[s][ks] = *reinterpret_cast<const half8_t*>(buf + s * HT + (w_rd ^ (ks * 32)));
  };
  auto mma = [&](auto sc, auto cc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, c = decltype(cc)::value;
#ifndef Y5_EMU
    __builtin_amdgcn_sched_barrier(0);
#ifndef Y5_G8_NOPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#endif
#ifdef Y5_G8_ABL_NOMMA
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      asm volatile("" :: "v"(wf[c][ks]));
#pragma unroll
      for (int pf = 0; pf < 2; ++pf) asm volatile("" :: "v"(af[pf][ks]));
    }
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pf = 0; pf < 2; ++pf)
        acc[s * 2 + pf][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[c][ks], af[pf][ks], acc[s * 2 + pf][c], 0, 0, 0);
#endif
#ifndef Y5_EMU
#ifndef Y5_G8_NOPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  auto lgkm0 = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0xC07F); };   // lgkmcnt(0), vmcnt untouched

  // ---- epilogue of the finished output tile j: bias + act (+ residual) -> two 16-byte stores per accumulator block, no LDS transpose ----
  T* yg = static_cast<T*>(p.y);   // may alias p.res
  const T* rg = static_cast<const T*>(p.res);
  T* y2g = static_cast<T*>(p.y2);
  // Stores (and residual loads) go through BUFFER addressing against per-tile resource descriptors (base = row m0 of the destination): a lane outside the layer
  // (m >= M, n >= C2) carries bit 31 in its offset and the hardware's range check drops it.  The pointer form spent ~20 of its 77 instructions per store on an
  // exec-mask branch and 64-bit address arithmetic -- the epilogue is instruction-issue bound (two waves per SIMD, 2 474 instructions each, section 4.2).
  const int e_rb = (wr * 128 + l31) * p.ldy * 2, e_rb2 = (wr * 128 + l31) * p.ld2 * 2, e_rbr = (wr * 128 + l31) * p.ldr * 2;
  auto epilogue = [&](int j, auto actc, auto modec) __attribute__((always_inline)) {
    constexpr bool ACT = decltype(actc)::value;   // (a per-element select on the runtime flag cost one more VALU per element)
    constexpr int MODE = decltype(modec)::value;  // 0: one destination; 1: + residual; 2: split store (C3's cv1 / cv2 halves) -- chosen once per launch
    int m0, n0;
    tile_coords(j, m0, n0);
    const y5_rsrc_t yrs = y5_make_rsrc(yg + (size_t)m0 * p.ldy, 0x7fffffffu);
    const y5_rsrc_t y2rs = MODE == 2 ? y5_make_rsrc(y2g + (size_t)m0 * p.ld2, 0x7fffffffu) : yrs;
    const y5_rsrc_t rrs = MODE == 1 ? y5_make_rsrc(rg + (size_t)m0 * p.ldr, 0x7fffffffu) : yrs;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nbase = n0 + wc * 64 + c * 32 + h * 16;   // wave-uniform
        const int n = nbase + g * 8;
        const int nb = n < p.Npad ? n : 0;
        const float4_t b0 = *reinterpret_cast<const float4_t*>(blds + nb), b1 = *reinterpret_cast<const float4_t*>(blds + nb + 4);
        const bool to2 = MODE == 2 && n >= p.split_n;
        (void)nbase;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int fr = (f >> 1) * 64 + (f & 1) * 32;
          const bool ok = m0 + wr * 128 + fr + l31 < p.M && n < p.C2;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = acc[f][c][h * 8 + e] + (e < 4 ? b0[e] : b1[e - 4]);
            v[e] = ACT ? y5_silu(t) : t;
          }
          uint4_t raw;
#pragma unroll
          for (int e = 0; e < 4; ++e) raw[e] = y5_pack_h2(v[2 * e], v[2 * e + 1]);
          if constexpr (MODE == 1) {   // (a layer with a residual has no split store)
            const uint4_t rr = y5_buffer_load16(rrs, ok ? e_rbr + fr * p.ldr * 2 + n * 2 : (int)0x80000000u, 0);
            half8_t a = __builtin_bit_cast(half8_t, raw), b = __builtin_bit_cast(half8_t, rr), o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)a[e] + (float)b[e]);
            raw = __builtin_bit_cast(uint4_t, o);
          }
          y5_buffer_store16(raw, yrs, ok && !to2 ? e_rb + fr * p.ldy * 2 + n * 2 : (int)0x80000000u, Y5_G8_ST_AUX);
          if constexpr (MODE == 2) y5_buffer_store16(raw, y2rs, ok && to2 ? e_rb2 + fr * p.ld2 * 2 + (n - p.split_n) * 2 : (int)0x80000000u, Y5_G8_ST_AUX);
        }
      }
    }
  };
  auto run_epilogue = [&](int j) __attribute__((always_inline)) {   // the launch's variant (wave-uniform launch parameters)
    using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>; using M2 = std::integral_constant<int, 2>;
    if (p.act) { if (p.split_n) epilogue(j, std::true_type{}, M2{}); else if (rg) epilogue(j, std::true_type{}, M1{}); else epilogue(j, std::true_type{}, M0{}); }
    else { if (p.split_n) epilogue(j, std::false_type{}, M2{}); else if (rg) epilogue(j, std::false_type{}, M1{}); else epilogue(j, std::false_type{}, M0{}); }
  };

  // ---- prologue: all of K tile 0, three half-tiles of K tile 1 ----
  loader_setup(0);
  tap_update();
  stage_wgt(I0{}, smem); stage_act(I0{}, smem); stage_wgt(I1{}, smem); stage_act(I1{}, smem);
  advance();
  stage_wgt(I0{}, smem + BUF); stage_act(I0{}, smem + BUF); stage_wgt(I1{}, smem + BUF);
#pragma unroll
  for (int q = 0; q < Gm::MAXN / 512; ++q) blds[tid + q * 512] = bv[q];
  advance_scalar();                            // K tile 2's offsets: the first q1 commits them
  y5_wait_vm<3 * Gm::LPH>();
  __builtin_amdgcn_s_waitcnt(0xC07F);          // the bias table is written before the barrier that publishes it
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the stagger: wave row 1 runs one barrier interval behind wave row 0
  Y5_G8_STAMP(2);
#ifdef Y5_G8_TIMING
  dbg_c0 = __builtin_amdgcn_s_memtime();
#endif

#ifdef Y5_G8_ABL_NODMA
  t_loop = true;
#endif
  int t = 0;
  for (int ti = 0; ti < nmine; ++ti) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][c][r] = 0.f;
    for (int kc = 0; kc < nk; ++kc, ++t) {
      char* cur = smem + (t & 1) * BUF;
      char* oth = smem + ((t & 1) ^ 1) * BUF;
      // q1
      rd_wgt(I0{}, cur);
#ifndef Y5_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      rd_act(I0{}, cur);
      stage_act(I1{}, oth);          // Act-h1 of K tile t+1
      advance_commit();              // the loader moves on to K tile t+2 (its scalars were computed in the previous q4)
      __builtin_amdgcn_s_waitcnt(0xC87F);   // lgkmcnt(8): the four filter reads have returned -- q2 restages Wgt-h0
      __builtin_amdgcn_s_barrier();
      lgkm0();
      mma(I0{}, I0{});
      __builtin_amdgcn_s_barrier();
      // q2
      rd_wgt(I1{}, cur);
      stage_wgt(I0{}, cur);          // Wgt-h0 of K tile t+2
      __builtin_amdgcn_s_barrier();
      lgkm0();
      mma(I0{}, I1{});
      __builtin_amdgcn_s_barrier();
      // q3
      rd_act(I1{}, cur);
      stage_act(I0{}, cur);          // Act-h0 of K tile t+2
      __builtin_amdgcn_s_barrier();
      lgkm0();
      mma(I1{}, I1{});
      __builtin_amdgcn_s_barrier();
#ifdef Y5_G8_ABL_NORD
      t_rd = true;
#endif
      // q4
      stage_wgt(I1{}, cur);          // Wgt-h1 of K tile t+2
      advance_scalar();              // K tile t+3's offsets, committed in the next q1
      y5_wait_vm<3 * Gm::LPH>();     // everything but the three youngest half-tiles has landed: K tile t+1 is complete
      __builtin_amdgcn_s_barrier();
      mma(I1{}, I0{});
      __builtin_amdgcn_s_barrier();
    }
#ifdef Y5_G8_TIMING
    if (ti == 0) { Y5_G8_STAMP(3); if (dbg_on) dbg[5] = __builtin_amdgcn_s_memtime() - dbg_c0; }
#endif
    // Tile boundary: the stagger is taken out for the epilogue and put back behind it.  Left in, wave row 1 sits at the barrier behind its last MFMAs until
    // wave row 0 has finished its WHOLE epilogue (wave row 0's next barrier is behind it), and wave row 0 then waits through wave row 1's: two epilogues
    // (3.9 us each at 7.Conv, profiles/r06/r06_g8_timing_v1.log) one after the other with the matrix pipe idle.  Wave row 0 takes its balancing barrier HERE
    // (it pairs with the barrier behind wave row 1's last MFMAs), both rows run their epilogues at the same time, and wave row 1 takes the extra barrier again
    // before the next tile's q1 (it pairs with the barrier behind wave row 0's q1 reads).  No LDS hazard moves: the epilogue touches only the bias table.
#ifndef Y5_G8_SERIAL_EPILOGUE
    if (wr == 0) __builtin_amdgcn_s_barrier();
#endif
    run_epilogue(ti);
#ifndef Y5_G8_SERIAL_EPILOGUE
    if (wr == 1 && ti + 1 < nmine) __builtin_amdgcn_s_barrier();
#endif
#ifdef Y5_G8_TIMING
    if (ti == 0) { Y5_G8_STAMP(4); if (dbg_on) { dbg[6] = nk; dbg[7] = nmine; } }
#endif
  }
#ifdef Y5_G8_SERIAL_EPILOGUE
  if (wr == 0) __builtin_amdgcn_s_barrier();   // balance the stagger
#endif
  Y5_DRAIN_VM();                                // the zero-fill LDS-DMA past the end must not outlive the wave
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// 256 pixels x 128 channels (id 96): the same structure for layers whose channel count / tile count does not fit 256-wide tiles (C2 = 128, 320, 640; M x N
// grids that leave the second round of 256 x 256 tiles half empty).  A wave owns 128 pixels x 32 channels (four accumulator blocks, 16 MFMAs per K tile);
// a K tile is THREE half-tiles (Act-h0, Act-h1, Wgt: all 128 filter rows = the 32 channels of the four wave columns) = 48 KiB, THREE K tiles are resident:
//         p1: read Wgt (4) then Act-h0 (8)      stage Act-h0 of K tile t+2                          MFMA act0 x wgt
//         p2: read Act-h1 (8)                   stage Act-h1 of K tile t+2, Wgt of K tile t+3       MFMA act1 x wgt        + s_waitcnt vmcnt(6)
// RAW: the p2 wait of K tile t leaves the three half-tiles staged in p1 / p2 of t in flight; K tile t+1 (Act staged in p1 / p2 of t-1, Wgt in p2 of t-2) is older.
// WAR: Act-h0 / Act-h1 of t+2 replace those of t-1 (buffer (t+2) % 3), read two phases earlier; Wgt of t+3 replaces Wgt of t, read first in p1 of t and retired
// by lgkmcnt(8) before p1's first barrier.  The filter walker therefore runs one K tile ahead of the activation walker.
struct Y5G8nGeom {
  static constexpr int NW = 8, BM = 256, BN = 128, BK = 64;
  static constexpr int HT = 128 * 128;
  static constexpr int BUF = 3 * HT;                // [Act-h0][Act-h1][Wgt]
  static constexpr int NB = 3;
  static constexpr int MAXN = 2048;
  static constexpr int OFF_BIAS = NB * BUF;
  static constexpr size_t LDS = OFF_BIAS + MAXN * 4;
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <bool UP2, bool SEQ = false, bool GEN = false>
__global__ __launch_bounds__(512, 2)
void y5_conv_g8n_kernel(const Y5ConvParams p) {
  static_assert(!GEN || (!UP2 && !SEQ), "the general-C1 loader serves plain layers in natural tap order");
  typedef half_t T;
  using Gm = Y5G8nGeom;
  constexpr int HT = Gm::HT, BUF = Gm::BUF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const blds = reinterpret_cast<float*>(smem + Gm::OFF_BIAS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 5, l31 = lane & 31;

  const int G = gridDim.x, bid = blockIdx.x;
  const int ntiles = p.tilesM * p.tilesN, nk = p.nk;
  const int nmine = (ntiles - bid + G - 1) / G;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t wrs = y5_make_rsrc(p.w, p.w_bytes);
  const y5_rsrc_t xrs2 = UP2 ? y5_make_rsrc(p.x2, p.x2_bytes) : xrs;

  float bv[Gm::MAXN / 512];
#pragma unroll
  for (int q = 0; q < Gm::MAXN / 512; ++q) bv[q] = tid + q * 512 < p.Npad ? p.bias[tid + q * 512] : 0.f;

  // ---- loaders: the activation walker (output tile a_t, chunk a_kc) and the filter walker (w_t, w_kc), the latter one K tile ahead ----
  const int lrow = lane >> 3, lslot = lane & 7;
  int a_base[2][2];
  int a_base2[UP2 ? 2 : 1][2];
  bool st_up = false;
  unsigned a_mask[2][2];
  unsigned w_off[2];
  int u_kh = 0, u_kw = 0, u_c0 = 0, u_ts = 0, a_t = 0, a_kc = 0, w_t = 0, w_kc = 0, w_ts = 0, w_c0 = 0;
  int st_tap_off = 0, st_tap_bit = 0;
  int st_off_hi = 0, st_nlo = 8;   // GEN (see y5_conv_g8_kernel)
  const int g_sl0 = lslot ^ ((((wave * 2 + 0) * 8 + lrow) >> 1) & 7), g_sl1 = lslot ^ ((((wave * 2 + 1) * 8 + lrow) >> 1) & 7);
  unsigned w_kcb = 0;   // byte offset of the filter walker's K tile inside a filter row
  const int kw_r = (256 + p.KW - 1) / p.KW;   // (see y5_conv_g8_kernel)
  auto tap_at = [&](int ts) __attribute__((always_inline)) {
    const int t = (int)((unsigned)(p.tap_seq >> (ts * 4)) & 15u);
    u_kh = (t * kw_r) >> 8;
    u_kw = t - u_kh * p.KW;
  };
  auto wgt_kcb = [&]() __attribute__((always_inline)) {
    const int t = (int)((unsigned)(p.tap_seq >> (w_ts * 4)) & 15u);
    w_kcb = (unsigned)((t * p.C1 + w_c0) * 2);
  };

  auto tile_coords = [&](int j, int& m0, int& n0) {
    const int t = y5_xcd_remap(bid + j * G, ntiles);
    const int tn = t % p.tilesN, tm = t / p.tilesN;
    m0 = tm * Gm::BM;
    n0 = tn * Gm::BN;
  };
  auto act_setup = [&](int j) __attribute__((always_inline)) {
    int m0, n0;
    tile_coords(j, m0, n0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int rr = (wave * 2 + jj) * 8 + lrow;
      const int sslot = lslot ^ ((rr >> 1) & 7);
#pragma unroll
      for (int s = 0; s < 2; ++s)
        y5_g8_act_row<UP2>(p, m0 + (rr >> 6) * 128 + s * 64 + (rr & 63), sslot, a_base[s][jj], a_mask[s][jj], UP2 ? &a_base2[s][jj] : nullptr);
    }
  };
  auto wgt_setup = [&](int j) __attribute__((always_inline)) {
    int m0, n0;
    tile_coords(j, m0, n0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int rr = (wave * 2 + jj) * 8 + lrow;
      const int sslot = lslot ^ ((rr >> 1) & 7);
      const int n = n0 + (rr >> 5) * 32 + Y5G8Geom::chan_of_row(rr & 31);
      w_off[jj] = n < p.Npad ? (unsigned)((n * p.Kpad + sslot * 8) * 2) : 0x80000000u;
    }
  };
  auto tap_update = [&]() {
    st_tap_off = ((u_kh * p.W + u_kw) * p.ldx + u_c0) * 2;
    st_tap_bit = u_kh * p.KW + u_kw;
    if constexpr (GEN) {
      st_nlo = (p.C1 - u_c0) >> 3;
      st_nlo = st_nlo < 8 ? st_nlo : 8;
      int kh1 = u_kh, kw1 = u_kw + 1;
      if (kw1 == p.KW) { kw1 = 0; ++kh1; }
      st_off_hi = ((kh1 * p.W + kw1) * p.ldx) * 2 - st_nlo * 16;
    }
    if constexpr (UP2) st_up = u_c0 < p.up_c;
  };
  auto adv_act = [&]() __attribute__((always_inline)) {
    u_c0 += 64;
    if constexpr (SEQ) {
      if (u_c0 >= p.C1 && a_kc + 1 < nk) { u_c0 = 0; tap_at(++u_ts); }
    } else {
      if (u_c0 >= p.C1) {
        u_c0 = GEN ? u_c0 - p.C1 : 0;
        if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
      }
    }
    if (++a_kc == nk) {
      a_kc = 0; u_kh = 0; u_kw = 0; u_c0 = 0; u_ts = 0;
      if constexpr (SEQ) tap_at(0);
      if (++a_t < nmine) act_setup(a_t);
      else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) a_mask[s][jj] = ~0u;
      }
    }
    tap_update();
  };
  auto adv_wgt = [&]() __attribute__((always_inline)) {
    if constexpr (SEQ) {
      w_c0 += 64;
      if (w_c0 >= p.C1) { w_c0 = 0; ++w_ts; }
    }
    if (++w_kc == nk) {
      w_kc = 0; w_ts = 0; w_c0 = 0;
      if (++w_t < nmine) wgt_setup(w_t);
      else { w_off[0] = 0x80000000u; w_off[1] = 0x80000000u; }
    }
    if constexpr (SEQ) wgt_kcb();
  };
  auto stage_act = [&](auto sc, char* buf) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      if constexpr (UP2) {
        if (st_up) {
          y5_bglds16(xrs2, (unsigned)(a_base2[s][jj] + st_tap_off) | (a_mask[s][jj] << 31), buf + s * HT + (wave * 2 + jj) * 1024);
          continue;
        }
      }
      unsigned voff;
      if constexpr (GEN) {
        const bool hi = (jj ? g_sl1 : g_sl0) >= st_nlo;
        voff = (unsigned)(a_base[s][jj] + (hi ? st_off_hi : st_tap_off)) | ((a_mask[s][jj] >> (hi ? st_tap_bit + 1 : st_tap_bit)) << 31);
      } else {
        voff = (unsigned)(a_base[s][jj] + st_tap_off) | ((a_mask[s][jj] >> st_tap_bit) << 31);
      }
      y5_bglds16(xrs, voff, buf + s * HT + (wave * 2 + jj) * 1024);
    }
  };
  auto stage_wgt = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) y5_bglds16(wrs, w_off[jj] + (SEQ ? w_kcb : (unsigned)(w_kc * 128)), buf + 2 * HT + (wave * 2 + jj) * 1024);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  const int fsw = (l31 >> 1) & 7;
  const int a_rd = (wr * 64 + l31) * 128 + ((g ^ fsw) << 4);
  const int w_rd = 2 * HT + (wc * 32 + l31) * 128 + ((g ^ fsw) << 4);
  half8_t af[2][4], wf[4];
  float16_t acc[4];

  auto rd_act = [&](auto sc, const char* buf) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
#pragma unroll
    for (int pf = 0; pf < 2; ++pf)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[pf][ks] = *reinterpret_cast<const half8_t*>(buf + s * HT + pf * 4096 + (a_rd ^ (ks * 32)));
  };
  auto rd_wgt = [&](const char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const half8_t*>(buf + (w_rd ^ (ks * 32)));
  };
  auto mma = [&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
#ifndef Y5_EMU
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pf = 0; pf < 2; ++pf) acc[s * 2 + pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], af[pf][ks], acc[s * 2 + pf], 0, 0, 0);
#ifndef Y5_EMU
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  auto lgkm0 = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0xC07F); };

  T* yg = static_cast<T*>(p.y);
  const T* rg = static_cast<const T*>(p.res);
  T* y2g = static_cast<T*>(p.y2);
  const int e_rb = (wr * 128 + l31) * p.ldy * 2, e_rb2 = (wr * 128 + l31) * p.ld2 * 2, e_rbr = (wr * 128 + l31) * p.ldr * 2;   // (buffer-addressed stores: see y5_conv_g8_kernel)
  auto epilogue = [&](int j, auto actc, auto modec) __attribute__((always_inline)) {
    constexpr bool ACT = decltype(actc)::value;
    constexpr int MODE = decltype(modec)::value;
    int m0, n0;
    tile_coords(j, m0, n0);
    const y5_rsrc_t yrs = y5_make_rsrc(yg + (size_t)m0 * p.ldy, 0x7fffffffu);
    const y5_rsrc_t y2rs = MODE == 2 ? y5_make_rsrc(y2g + (size_t)m0 * p.ld2, 0x7fffffffu) : yrs;
    const y5_rsrc_t rrs = MODE == 1 ? y5_make_rsrc(rg + (size_t)m0 * p.ldr, 0x7fffffffu) : yrs;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int nbase = n0 + wc * 32 + h * 16;
      const int n = nbase + g * 8;
      const int nb = n < p.Npad ? n : 0;
      const float4_t b0 = *reinterpret_cast<const float4_t*>(blds + nb), b1 = *reinterpret_cast<const float4_t*>(blds + nb + 4);
      const bool to2 = MODE == 2 && n >= p.split_n;
      (void)nbase;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int fr = (f >> 1) * 64 + (f & 1) * 32;
        const bool ok = m0 + wr * 128 + fr + l31 < p.M && n < p.C2;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = acc[f][h * 8 + e] + (e < 4 ? b0[e] : b1[e - 4]);
          v[e] = ACT ? y5_silu(t) : t;
        }
        uint4_t raw;
#pragma unroll
        for (int e = 0; e < 4; ++e) raw[e] = y5_pack_h2(v[2 * e], v[2 * e + 1]);
        if constexpr (MODE == 1) {
          const uint4_t rr = y5_buffer_load16(rrs, ok ? e_rbr + fr * p.ldr * 2 + n * 2 : (int)0x80000000u, 0);
          half8_t a = __builtin_bit_cast(half8_t, raw), b = __builtin_bit_cast(half8_t, rr), o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)a[e] + (float)b[e]);
          raw = __builtin_bit_cast(uint4_t, o);
        }
        y5_buffer_store16(raw, yrs, ok && !to2 ? e_rb + fr * p.ldy * 2 + n * 2 : (int)0x80000000u, Y5_G8_ST_AUX);
        if constexpr (MODE == 2) y5_buffer_store16(raw, y2rs, ok && to2 ? e_rb2 + fr * p.ld2 * 2 + (n - p.split_n) * 2 : (int)0x80000000u, Y5_G8_ST_AUX);
      }
    }
  };
  auto run_epilogue = [&](int j) __attribute__((always_inline)) {
    using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>; using M2 = std::integral_constant<int, 2>;
    if (p.act) { if (p.split_n) epilogue(j, std::true_type{}, M2{}); else if (rg) epilogue(j, std::true_type{}, M1{}); else epilogue(j, std::true_type{}, M0{}); }
    else { if (p.split_n) epilogue(j, std::false_type{}, M2{}); else if (rg) epilogue(j, std::false_type{}, M1{}); else epilogue(j, std::false_type{}, M0{}); }
  };

  // ---- prologue: K tiles 0 and 1 complete, the filter half-tile of K tile 2 ----
  act_setup(0);
  wgt_setup(0);
  if constexpr (SEQ) tap_at(0);
  tap_update();
  if constexpr (SEQ) wgt_kcb();
  stage_wgt(smem); adv_wgt(); stage_act(I0{}, smem); stage_act(I1{}, smem); adv_act();
  stage_wgt(smem + BUF); adv_wgt(); stage_act(I0{}, smem + BUF); stage_act(I1{}, smem + BUF); adv_act();
  stage_wgt(smem + 2 * BUF); adv_wgt();
#pragma unroll
  for (int q = 0; q < Gm::MAXN / 512; ++q) blds[tid + q * 512] = bv[q];
  y5_wait_vm<8>();   // K tile 0 (the six oldest loads) has landed
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();

  char* b_cur = smem;             // buffer of K tile t
  char* b_nx1 = smem + BUF;       // t + 1
  char* b_nx2 = smem + 2 * BUF;   // t + 2
  for (int ti = 0; ti < nmine; ++ti) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    for (int kc = 0; kc < nk; ++kc) {
      // p1
      rd_wgt(b_cur);
#ifndef Y5_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      rd_act(I0{}, b_cur);
      stage_act(I0{}, b_nx2);        // Act-h0 of K tile t+2
      __builtin_amdgcn_s_waitcnt(0xC87F);   // lgkmcnt(8): the four filter reads have returned -- p2 restages Wgt
      __builtin_amdgcn_s_barrier();
      lgkm0();
      mma(I0{});
      __builtin_amdgcn_s_barrier();
      // p2
      rd_act(I1{}, b_cur);
      stage_act(I1{}, b_nx2);        // Act-h1 of K tile t+2
      adv_act();
      stage_wgt(b_cur);              // Wgt of K tile t+3 (buffer t % 3)
      adv_wgt();
      y5_wait_vm<6>();               // K tile t+1 is complete
      __builtin_amdgcn_s_barrier();
      lgkm0();
      mma(I1{});
      __builtin_amdgcn_s_barrier();
      char* const b = b_cur; b_cur = b_nx1; b_nx1 = b_nx2; b_nx2 = b;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // (tile boundary: see y5_conv_g8_kernel)
    run_epilogue(ti);
    if (wr == 1 && ti + 1 < nmine) __builtin_amdgcn_s_barrier();
  }
  Y5_DRAIN_VM();
}
