// Weight gradient of the 3x3 convolutions (k3 p1, stride 1 or 2 -- every 3x3 of the yolov5 graphs; autograd of models/common.py:74-92 `Conv`)
// with the activation staged as SPATIAL PATCHES instead of gathered im2col rows.
//
// The general kernel (wgrad.hip) tiles k = (kh, kw, c) in 64/128-wide slices and gathers one 32-pixel im2col chunk per slice: over all
// slices every activation pixel travels L2 -> LDS nine times and every dz row once per k slice (128 -> 128 k3: 9 + 9 units of traffic
// for 1 + 1 units of tensor).  The round-2 per-layer table (profiles/r02/r02_wgrad_splits.log) has exactly these layers at 3-6.5x their
// HBM time while the pointwise layers sit at 1.1-1.5x.  Here ONE workgroup owns all nine taps of a (32 NT) x (32 CT) channel tile:
//     chunk   = two SEGMENTS of 16 consecutive output pixels of one output row (16 = the contraction depth of one MFMA 32x32x16)
//     staged  = dz[2][16 px][32 NT ch]  +  x patch [2][3 rows][15 s + 3 px][32 CT ch]   (LDS-DMA, out-of-image / tail = zero fill)
//     waves   = 9, wave t owns tap (kh, kw) = (t / 3, t % 3): NT x CT accumulators, its B fragments are the patch shifted by the tap
// so the activation is staged 3 (1 + 2/16) times (the three patch rows of neighbouring output rows overlap), dz once per c tile.
// Segment coordinates (b, oh, ow0) are workgroup-uniform and advance incrementally in scalar registers; a staged 16-byte piece costs
// two adds, two range tests and a select.  Fragments are transposed reads (8 pixels of one channel per lane), gathered with 16-bit
// LDS reads as in the general kernel.  Pixel-range splits over the grid; partial sums leave by atomics or (DET) through per-split slabs.
#pragma once

// One MFMA fragment (lane (fi, g): pixels 8 g .. 8 g + 7 of channel fi) from a pixel-major LDS tile with gfx950's transpose read
// ds_read_b64_tr_b16 (semantics pinned on the hardware by scripts/ubench/tr_probe.hip, modelled in tests/hipemu): within a 16-lane group, lane i
// receives element i % 4 of the 8 bytes addressed by lanes 4 j + i / 4 (j = 0..3).  With lane (i, grp) addressing row i / 4 (+ 8 (grp >> 1)),
// bytes 8 (i % 4) .. + 7 of the 32-byte half grp & 1 of a 32-channel block, every lane ends up with FOUR CONSECUTIVE ROWS of column lane & 31:
// two reads per fragment instead of eight 16-bit reads + four v_perm (the first version of this kernel issued 6 LDS reads per MFMA and its
// phases -- staging, fragment reads, MFMAs -- ran one after the other behind the per-chunk barrier: profiles/r03/r03_wgrad3_ablation.txt).
typedef short y5_s4_t __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ half8_t y5_tr_frag(const char* rows0to3, int rows4to7_off) {
  struct { y5_s4_t lo, hi; } f;
  f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) y5_s4_t*)(rows0to3));
  f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) y5_s4_t*)(rows0to3 + rows4to7_off));
  return __builtin_bit_cast(half8_t, f);
}
// Bank-conflict-free tile rows for those reads: a read touches four consecutive rows x 64 bytes; with 128- / 256-byte rows they would share banks,
// so the 64-byte block b of row r lives at block b ^ swz(r) -- a permutation the staging applies for free (every LDS-DMA lane picks its own
// global address).  ROWB = 64: rows 0..3 already cover the 64 banks.
template <int ROWB> __device__ __forceinline__ int y5_wg3_swz(int r) { return ROWB == 256 ? (r & 3) : ROWB == 128 ? ((r >> 1) & 1) : 0; }

template <int NT, int CT, int STR, int S, bool DET>
__global__ __launch_bounds__(576)
void y5_conv_wgrad3_kernel(const Y5WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PWp = 15 * STR + 3;                  // patch pixels per row: iw = ow*STR - 1 + kw, ow = 0..15, kw = 0..2
  constexpr int ZROW = 64 * NT, XROW = 64 * CT;      // bytes per staged pixel
  constexpr int ZT = 32 * ZROW;
  constexpr int XPIX = 2 * 3 * PWp;
  constexpr int ZI = ZT / 1024, XI = (XPIX * XROW + 1023) / 1024;
  constexpr int XT = XI * 1024, BUF = ZT + XT;
  constexpr int NI = (ZI + XI + 8) / 9;              // LDS-DMA instructions per wave and chunk (slot q of wave w = instruction 9 q + w)
  constexpr int LPZ = 4 * NT, LPX = 4 * CT;          // lanes (16 bytes each) per pixel
  constexpr int HALFP = (PWp + 1) / 2;               // stride 2: even patch pixels 0, 2, .. first (HALFP of them), then the odd ones
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  const int tc = bid % p.tiles_k;
  const int tn = (bid / p.tiles_k) % p.tiles_n;
  const int sp = bid / (p.tiles_k * p.tiles_n);
  const int n0 = tn * 32 * NT, c0 = tc * 32 * CT;
  const int spr = (p.OW + 15) >> 4;                   // segments per output row
  const int total_segs = p.B * p.OH * spr;
  const int seg_begin = sp * p.pix_per_split;         // (pix_per_split = SEGMENTS per split here, even)
  const int seg_end = seg_begin + p.pix_per_split < total_segs ? seg_begin + p.pix_per_split : total_segs;
  if (seg_begin >= seg_end) return;
  const int nchunks = (seg_end - seg_begin + 1) >> 1;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t zrs = y5_make_rsrc(p.dz, p.dz_bytes);

  // workgroup-uniform coordinates of the two segments staged NEXT
  int s_idx[2], s_b[2], s_oh[2], s_sg[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    s_idx[t] = seg_begin + t;
    const int row = s_idx[t] / spr;
    s_sg[t] = s_idx[t] - row * spr;
    s_b[t] = row / p.OH;
    s_oh[t] = row - s_b[t] * p.OH;
  }

  // per-lane constants of this wave's staging slots, packed (the 4 x 2 tile keeps 128 accumulators: every register counts):
  //   meta = seg | r << 1 | px << 3 | ok << 9 | real << 10   (seg: which of the chunk's two segments; r, px: patch row / pixel; real: not a dummy lane)
  bool i_isx[NI];      // wave-uniform: which buffer resource the slot's instruction uses
  int i_dst[NI];       // wave-uniform: LDS offset of the instruction's 1 KiB
  int i_meta[NI], i_const[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int I = q * 9 + wave;
    i_isx[q] = I >= ZI;
    i_dst[q] = I < ZI ? I * 1024 : ZT + (I - ZI) * 1024;
    i_meta[q] = 0; i_const[q] = 0;
    if (I < ZI) {
      const int pr = I * (64 / LPZ) + lane / LPZ, pch = lane % LPZ;                       // physical row (pixel of the chunk), 16-byte chunk of the row
      const int cg = (((pch >> 2) ^ y5_wg3_swz<ZROW>(pr)) << 2) | (pch & 3);               // logical channel group stored there
      const int px = pr & 15;
      i_meta[q] = (pr >> 4) | px << 3 | (n0 + 8 * cg < p.C2 ? 1 << 9 : 0) | 1 << 10;
      i_const[q] = (px * p.ldz + n0 + 8 * cg) * 2;
    } else if (I < ZI + XI) {
      const int t = (I - ZI) * (64 / LPX) + lane / LPX, pch = lane % LPX;                  // physical row = patch position, chunk
      if (t < XPIX) {
        const int cg = (((pch >> 2) ^ y5_wg3_swz<XROW>(t)) << 2) | (pch & 3);
        const int sg = t / (3 * PWp), rem = t - sg * (3 * PWp);
        const int r = rem / PWp, pc = rem - r * PWp;
        // stride 2: a patch row is stored EVEN pixels first, then odd ones, so that the pixels 2 px + kw of a tap are consecutive rows
        const int px = STR == 1 ? pc : pc < HALFP ? 2 * pc : 2 * (pc - HALFP) + 1;
        i_meta[q] = sg | r << 1 | px << 3 | (c0 + 8 * cg < p.C1 ? 1 << 9 : 0) | 1 << 10;
        i_const[q] = ((r * p.W + px) * p.ldx + c0 + 8 * cg) * 2;
      }
    }
  }

  // a wave whose last slot lies past the ZI + XI instructions of a chunk issues NI - 1 per chunk (its vmcnt bookkeeping below follows)
  const bool short_wave = (NI - 1) * 9 + wave >= ZI + XI;

  auto stage = [&](int buf) {
    char* base = smem + buf * BUF;
    // uniform bases of the two segments: dz row start, x patch origin (ih = oh*STR - 1, iw = ow0*STR - 1; may be "negative": only used when in range)
    int zb[2], xb[2], ih0[2], iw0[2], ow0[2];
    bool live[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      live[t] = s_idx[t] < seg_end;
      ow0[t] = s_sg[t] << 4;
      zb[t] = (((s_b[t] * p.OH + s_oh[t]) * p.OW + ow0[t]) * p.ldz) * 2;
      ih0[t] = s_oh[t] * STR - 1;
      iw0[t] = ow0[t] * STR - 1;
      xb[t] = (((s_b[t] * p.H + ih0[t]) * p.W + iw0[t]) * p.ldx) * 2;
    }
#pragma unroll
    for (int q = 0; q < NI; ++q) {   // branch-free per lane: range tests are bit operations, the offset a select
      const int m = i_meta[q];
      const bool t = m & 1;
      const int px = (m >> 3) & 63;
      bool ok = ((m >> 9) == 3) & (t ? live[1] : live[0]);   // real lane, channel group inside the tensor, segment inside the split
      int off;
      if (!i_isx[q]) {                                       // (wave-uniform)
        ok &= (t ? ow0[1] : ow0[0]) + px < p.OW;
        off = (t ? zb[1] : zb[0]) + i_const[q];
      } else {
        const int ih = (t ? ih0[1] : ih0[0]) + ((m >> 1) & 3), iw = (t ? iw0[1] : iw0[0]) + px;
        ok &= ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W);
        off = (t ? xb[1] : xb[0]) + i_const[q];
      }
#ifdef Y5_WG_NOSTAGE
      ok &= off == 0x12345;
      if (__builtin_amdgcn_ballot_w64(ok))
#endif
      if (q < NI - 1 || !short_wave) y5_bglds16(i_isx[q] ? xrs : zrs, ok ? (unsigned)off : Y5_OOB, base + i_dst[q]);   // (wave-uniform)
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {   // advance both segment slots by one chunk = two segments: at most two row wraps (spr = 1)
      s_idx[t] += 2;
      s_sg[t] += 2;
#pragma unroll
      for (int w = 0; w < 2; ++w)
        if (s_sg[t] >= spr) {
          s_sg[t] -= spr;
          if (++s_oh[t] == p.OH) { s_oh[t] = 0; ++s_b[t]; }
        }
    }
  };

  const int kh = wave / 3, kw = wave - kh * 3;
  const int fi = lane & 31, g = lane >> 5;
  // transpose-read roles (y5_tr_frag): lane (i = lane & 15, grp = lane >> 4) addresses row 8 (grp >> 1) + i / 4, bytes 32 (grp & 1) + 8 (i % 4) of a block
  const int q4 = (lane & 15) >> 2;
  const int tr_col = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const int tr_z = (8 * g + q4) * ZROW + tr_col, tr_x = (8 * g + q4) * XROW + tr_col;
  const int tap_row = kh * PWp + (STR == 1 ? kw : (kw & 1) * HALFP + (kw >> 1));   // (uniform) patch row of the tap's pixel for output pixel 0
  float16_t acc[NT][CT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b2 = 0; b2 < CT; ++b2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.f;

#pragma unroll
  for (int s = 0; s < S - 1; ++s) stage(s);
  int cur = 0, nxt = S - 1;
  for (int ch = 0; ch < nchunks; ++ch) {
    if (short_wave) y5_wait_vm<(S - 2) * (NI - 1)>();
    else y5_wait_vm<(S - 2) * NI>();
    __builtin_amdgcn_s_barrier();
    stage(nxt);
    const char* zt = smem + cur * BUF + tr_z;          // lane's row / 8-byte piece of the dz tile
    const char* xt = smem + cur * BUF + ZT + tr_x;     // ... of the patch, before the tap / segment row offset
    nxt = cur;
    cur = cur + 1 == S ? 0 : cur + 1;
#pragma unroll NT * CT >= 8 ? 1 : 2   // the 4 x 2 tile: 128 accumulators + both segments' fragments in flight do not fit 168 registers
    for (int ks = 0; ks < 2; ++ks) {
      half8_t af[NT], bf[CT];
      const int t0 = ks * 3 * PWp + tap_row;           // (uniform) first patch row of the tap in segment ks
#ifndef Y5_WG_NOREAD
#pragma unroll
      for (int b2 = 0; b2 < CT; ++b2) bf[b2] = y5_tr_frag(xt + t0 * XROW + ((b2 ^ y5_wg3_swz<XROW>(t0 + q4)) << 6), 4 * XROW);
#pragma unroll
      for (int a = 0; a < NT; ++a) af[a] = y5_tr_frag(zt + ks * 16 * ZROW + ((a ^ y5_wg3_swz<ZROW>(q4)) << 6), 4 * ZROW);
#else
#pragma unroll
      for (int b2 = 0; b2 < CT; ++b2) bf[b2] = __builtin_bit_cast(half8_t, uint4_t{(uint32_t)lane, (uint32_t)ch, 5u, (uint32_t)b2});
#pragma unroll
      for (int a = 0; a < NT; ++a) af[a] = __builtin_bit_cast(half8_t, uint4_t{(uint32_t)ch, (uint32_t)lane, 3u, (uint32_t)a});
#endif
#pragma unroll
      for (int a = 0; a < NT; ++a) {
#ifndef Y5_WG_NOMFMA
#pragma unroll
        for (int b2 = 0; b2 < CT; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b2], acc[a][b2], 0, 0, 0);
#else
#pragma unroll
        for (int b2 = 0; b2 < CT; ++b2)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[a][b2][e] += (float)af[a][e] * (float)bf[b2][e];
#endif
      }
    }
  }
  y5_wait_vm<0>();
  // D[i][j]: col j = lane & 31 (c), row i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (n); stores / atomics are coalesced along k (lanes = c)
#pragma unroll
  for (int b2 = 0; b2 < CT; ++b2) {
    const int c = c0 + b2 * 32 + fi;
    if (c >= p.C1) continue;
    const int kcol = wave * p.C1 + c;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (n >= p.C2) continue;
#ifdef Y5_WG_NOATOM
        if (acc[a][b2][r] == 1.2345f) p.dw[(size_t)n * p.Kpad + kcol] = acc[a][b2][r];
        continue;
#endif
        if constexpr (DET) p.ws[((size_t)sp * p.Npad + n) * p.Kpad + kcol] = acc[a][b2][r];
        else atomicAdd(p.dw + (size_t)n * p.Kpad + kcol, acc[a][b2][r]);
      }
  }
}

// ---- weight gradient of the STEM (0.Conv: k6 s2 p2 on 3 channels, run as k(6,3) s(2,1) p(2,1) on the paired-pixel view with 8 channels) ----------------
// The general kernel gathers 18 taps x 8 channels per pixel for this layer: 533 us per step against 108 us of HBM time (420 MB of dz, 210 MB of x) --
// one seventh of the whole weight-gradient time (profiles/r03/r03_train_kernel_stats_v1.csv).  Here one workgroup of SIX waves owns the filter
// (32 x 144): wave kh multiplies filter row kh.  In the paired view the 24 k columns (kw, c) of a filter row are 48 contiguous bytes of x starting at
// pixel ow - 1, so the B tile of output pixel p is the 64-byte window at patch pixel p (columns 24..31 are dead and masked at the end): a transposed
// fragment is two ds_read_b64_tr_b16 over rows that are 16 bytes apart.  A chunk is NSEG segments of 16 output pixels of one output row:
// dz[NSEG][16][32] + patch[NSEG][6 rows][20 px][8], staged by LDS-DMA through an S-deep ring exactly like the kernel above.
template <int NSEG, int S, bool DET>
__global__ __launch_bounds__(384)
void y5_conv_wgrad_stem_kernel(const Y5WgradParams p) {
  static_assert(NSEG <= 8, "segment index: 3 bits of the staging meta word");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PWs = 20;                            // patch pixels per row: ow0 - 1 .. ow0 + 18 (16 windows of 4 pixels)
  constexpr int ZT = NSEG * 16 * 64;
  constexpr int XPIX = NSEG * 6 * PWs;
  constexpr int ZI = ZT / 1024, XI = (XPIX * 16 + 1023) / 1024;
  constexpr int XT = XI * 1024 + 64, BUF = ZT + XT;   // (+64: the last window of the last row reads 48 bytes past its 16)
  constexpr int NI = (ZI + XI + 5) / 6;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = kh
  const int bid = blockIdx.x;
  const int tn = bid % p.tiles_n, sp = bid / p.tiles_n;
  const int n0 = tn * 32;
  const int spr = (p.OW + 15) >> 4;
  const int total_segs = p.B * p.OH * spr;
  const int seg_begin = sp * p.pix_per_split;                  // segments per split, a multiple of NSEG
  const int seg_end = seg_begin + p.pix_per_split < total_segs ? seg_begin + p.pix_per_split : total_segs;
  if (seg_begin >= seg_end) return;
  const int nchunks = (seg_end - seg_begin + NSEG - 1) / NSEG;
  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t zrs = y5_make_rsrc(p.dz, p.dz_bytes);

  // (uniform) first segment of the chunk staged next
  int c_idx = seg_begin;
  int c_row = c_idx / spr, c_sg = c_idx - c_row * spr;
  int c_b = c_row / p.OH, c_oh = c_row - c_b * p.OH;

  // per-lane constants: meta = seg (3 bits) | r << 3 | u << 6 | ok << 11 | real << 12
  bool i_isx[NI];
  int i_dst[NI], i_meta[NI], i_const[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int I = q * 6 + wave;
    i_isx[q] = I >= ZI;
    i_dst[q] = I < ZI ? I * 1024 : ZT + (I - ZI) * 1024;
    i_meta[q] = 0; i_const[q] = 0;
    if (I < ZI) {
      const int pr = I * 16 + (lane >> 2), cg = lane & 3;      // 64-byte rows: 16 pixels per instruction; no swizzle needed (y5_wg3_swz<64>)
      const int px = pr & 15;
      i_meta[q] = (pr >> 4) | px << 6 | (n0 + 8 * cg < p.C2 ? 1 << 11 : 0) | 1 << 12;
      i_const[q] = (px * p.ldz + n0 + 8 * cg) * 2;
    } else if (I < ZI + XI) {
      const int t = (I - ZI) * 64 + lane;                      // one 16-byte patch pixel per lane
      if (t < XPIX) {
        const int sg = t / (6 * PWs), rem = t - sg * (6 * PWs);
        const int r = rem / PWs, u = rem - r * PWs;
        i_meta[q] = sg | r << 3 | u << 6 | 1 << 11 | 1 << 12;
        i_const[q] = ((r * p.W + u) * p.ldx) * 2;
      }
    }
  }
  const bool short_wave = (NI - 1) * 6 + wave >= ZI + XI;

  auto stage = [&](int buf) {
    char* base = smem + buf * BUF;
    int zb[NSEG], xb[NSEG], ih0[NSEG], iw0[NSEG], ow0[NSEG];
    bool live[NSEG];
    {
      int b = c_b, oh = c_oh, sg = c_sg;
#pragma unroll
      for (int t = 0; t < NSEG; ++t) {
        live[t] = c_idx + t < seg_end;
        ow0[t] = sg << 4;
        zb[t] = (((b * p.OH + oh) * p.OW + ow0[t]) * p.ldz) * 2;
        ih0[t] = oh * 2 - 2;
        iw0[t] = ow0[t] - 1;
        xb[t] = (((b * p.H + ih0[t]) * p.W + iw0[t]) * p.ldx) * 2;
        if (++sg == spr) { sg = 0; if (++oh == p.OH) { oh = 0; ++b; } }
      }
      c_idx += NSEG; c_b = b; c_oh = oh; c_sg = sg;            // the state after NSEG single steps = the next chunk's first segment
    }
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int m = i_meta[q];
      const int t = m & 7, u = (m >> 6) & 31;
      bool lv = live[0]; int zbt = zb[0], xbt = xb[0], ih = ih0[0], iw = iw0[0], owt = ow0[0];
#pragma unroll
      for (int k = 1; k < NSEG; ++k)
        if (t == k) { lv = live[k]; zbt = zb[k]; xbt = xb[k]; ih = ih0[k]; iw = iw0[k]; owt = ow0[k]; }
      bool ok = ((m >> 11) == 3) & lv;
      int off;
      if (!i_isx[q]) {
        ok &= owt + u < p.OW;
        off = zbt + i_const[q];
      } else {
        ih += (m >> 3) & 7; iw += u;
        ok &= ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W);
        off = xbt + i_const[q];
      }
      if (q < NI - 1 || !short_wave) y5_bglds16(i_isx[q] ? xrs : zrs, ok ? (unsigned)off : Y5_OOB, base + i_dst[q]);
    }
  };

  const int fi = lane & 31, g = lane >> 5;
  const int q4 = (lane & 15) >> 2;
  const int tr_col = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const int tr_z = (8 * g + q4) * 64 + tr_col;                 // dz tile: 64-byte rows
  const int tr_x = (8 * g + q4) * 16 + tr_col;                 // patch: the window of pixel p starts 16 p bytes into the row
  float16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

#pragma unroll
  for (int s = 0; s < S - 1; ++s) stage(s);
  int cur = 0, nxt = S - 1;
  for (int ch = 0; ch < nchunks; ++ch) {
    if (short_wave) y5_wait_vm<(S - 2) * (NI - 1)>();
    else y5_wait_vm<(S - 2) * NI>();
    __builtin_amdgcn_s_barrier();
    stage(nxt);
    const char* zt = smem + cur * BUF + tr_z;
    const char* xt = smem + cur * BUF + ZT + wave * (PWs * 16) + tr_x;
    nxt = cur;
    cur = cur + 1 == S ? 0 : cur + 1;
#pragma unroll
    for (int ks = 0; ks < NSEG; ++ks) {
      const half8_t bf = y5_tr_frag(xt + ks * (6 * PWs * 16), 4 * 16);
      const half8_t af = y5_tr_frag(zt + ks * (16 * 64), 4 * 64);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
  }
  y5_wait_vm<0>();
  if (fi >= 24) return;                                        // dead columns of the 64-byte window
  const int kcol = wave * 24 + fi;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * g;
    if (n >= p.C2) continue;
    if constexpr (DET) p.ws[((size_t)sp * p.Npad + n) * p.Kpad + kcol] = acc[r];
    else atomicAdd(p.dw + (size_t)n * p.Kpad + kcol, acc[r]);
  }
}
