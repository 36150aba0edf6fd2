// NHWC implicit-GEMM convolution for gfx950 (MI355X): MFMA 32x32 tiles, LDS-DMA staged input/filter tiles
// (global_load_lds, 16 B/lane, XOR-swizzled on the SOURCE side), PERSISTENT workgroups that walk several output
// tiles with the next tile's loads in flight while the previous tile's epilogue runs, and a fused
// bias + SiLU + residual epilogue that is transposed through a per-wave LDS scratch so that every global
// store (and the optional 2x-upsampled replica) is a 16-byte, row-contiguous access.
//
// Replaces, for the reference hot path, `Conv.forward_fuse` (models/common.py:90-92: act(conv(x)) with the
// BN folded by utils/torch_utils.py:224-254), the residual add of `Bottleneck.forward` (common.py:181),
// the channel concat of `C3`/`SPPF`/`Concat` (common.py:246,340,453: the output is written into a channel
// slice [c_off, c_off+C2) of a wider NHWC buffer, pixel stride ldy) and `nn.Upsample(2,'nearest')`
// (yolov5s.yaml:36,41: optional second, 2x-replicated store).
//
// GEMM view (operands swapped so that one lane owns ONE output pixel and 4 consecutive channels):
//     D[n][m] = sum_k  Wp[n][k] * A[m][k]      n = output channel, m = output pixel (b,oh,ow),
//     k = (kh, kw, c) flattened, A gathered on the fly from x[b][oh*SH-PH+kh][ow*SW-PW+kw][c].
// MFMA "A" operand = filter rows, "B" operand = activation rows; both are RB-byte LDS rows (RB = 64 or 128:
// BK = 32/64 halfs or 16 floats of K) read with ds_read_b128: lane l -> row (l & 31), 16-byte slot from (l >> 5).
// Any permutation of k inside a row is legal as long as filter and activation use the same one.
//
// LDS row swizzle: 16-byte slot s of row r is stored at slot s ^ f(r), f(r) = (r>>2)&3 for 64-byte rows and
// (r>>1)&7 for 128-byte rows; the four 16-lane groups of a ds_read_b128 then hit 16 distinct 16-byte bank slots.
// The LDS-DMA destination must stay lane-linear, so the permutation is applied to the per-lane GLOBAL source.
#pragma once
#include "y5_common.h"


#ifdef Y5_DBG_TIMING
__device__ unsigned long long y5_dbg_timing[128];
__device__ unsigned long long y5_dbg_blocks[4096];  // per workgroup: kernel entry, loop start, loop end, exit (s_memrealtime, 100 MHz)
#endif

struct Y5ConvParams {
  const void* x;      // input  NHWC, pixel stride ldx elements
  const void* w;      // packed filter [Npad][Kpad], k = (kh,kw,c)
  const float* bias;  // [Npad] fp32
  const void* res;    // optional residual, same geometry as y (pixel stride ldr), may alias y
  void* y;            // output NHWC slice, pixel stride ldy (may be null when only y2 is wanted)
  void* y2;           // optional second destination: 2x nearest-upsampled copy (pixel stride ld2)
  const void* zero;   // >= 64 bytes of zeros in global memory (pointer-addressed kernels: padding taps / tails)
  unsigned x_bytes, w_bytes;  // extents of x / w for the buffer resource descriptors (offsets beyond them read as zeros)
  int o_mul_h, o_mul_w, o_off_h, o_off_w, o_H, o_W;  // o_mul_h != 0: output pixel (b,oh,ow) is stored at
                                                       // (b, oh*o_mul_h + o_off_h, ow*o_mul_w + o_off_w) of an o_H x o_W image (dgrad parity classes)
  int B, H, W, C1, ldx;
  int OH, OW, C2, ldy;
  int KH, KW, SH, SW, PH, PW;
  int act;  // 0 = identity, 1 = SiLU
  int Kpad, Npad, K;
  int ldr, ld2;
  int M;  // B*OH*OW
  int tilesM, tilesN, nk;
  float* sk_ws;        // stream-K (SK kernels): one fp32 partial-tile slab per workgroup, [G][WM*WN*TM*TN*16*64]
  unsigned* sk_flags;  // [G]: 1 = workgroup g's slab holds the partial sums of the tile its unit range starts in; reset to 0 by the reader
  // conv_k3.h with a fused 1x1 behind the 3x3 (PW2): second filter [pw2_npad][pw2_kpad] (k = the 3x3's output channels), fp32 bias, activation,
  // output channels [0, pw2_split) -> y (pixel stride ldy), [pw2_split, pw2_c2) -> y2 (pixel stride ld2); pw2_split == pw2_c2: everything to y
  const void* pw2_w;
  const float* pw2_bias;
  unsigned pw2_w_bytes;
  int pw2_kpad, pw2_npad, pw2_c2, pw2_act, pw2_split;
  int h3_th, h3_tw, h3_tiles_h, h3_tiles_w;  // conv_h3.h: spatial tile (output rows x columns) and tiles per image
  int split_n;  // > 0: output channels >= split_n go to y2 (pixel stride ld2, channel n - split_n) instead of y -- C3's cv1 / cv2 halves
                // of one GEMM landing in two different buffers (y2 is then NOT the upsampled replica)
  // UP2 kernels (1x1 s1 p0 only): `nn.Upsample(2, 'nearest')` + `Concat` consumed VIRTUALLY (models/yolov5s.yaml:36-37,41-42, common.py:443-453).
  // Input channels [0, up_c) of output pixel (b, oh, ow) are read from the LOW-resolution tensor x2 (B, H/2, W/2, >= up_c channels, pixel
  // stride ldx2) at (b, oh >> 1, ow >> 1); channels [up_c, C1) from x as usual (x = the concat buffer whose first up_c channels are never written).
  const void* x2;
  unsigned x2_bytes;
  int ldx2, up_c;
  // exact unsigned division by OH*OW and by OW as multiply-high + shifts (Granlund-Montgomery, filled by y5_conv_set_fastdiv on the host): the
  // loader decomposes every staged row's pixel index once per tile -- two ~40-instruction division sequences per row otherwise
  unsigned dv_ohw_m, dv_ow_m;
  int dv_ohw_s, dv_ow_s;   // sh1 | sh2 << 8
  // y5_conv2d_fwd_stats (train-mode forward, act = 0, streaming kernels): one row [2][C2] of per-channel (sum z, sum z^2) per workgroup (y5_common.h Y5StatAcc)
  float* bn_partial;
  size_t bn_bytes;         // host side: capacity of bn_partial
  int* bn_rows;            // host side: receives the number of rows (= grid size) of this launch
  // conv_g8.h: the ORDER in which the K loop walks the filter taps -- 4 bits per position, position i holds the tap kh * KW + kw staged i-th; 0 = natural
  // order.  Stride-2 3x3 layers walk the taps grouped by the input-pixel class they touch ((odd row, odd column): four taps, ...), so that a cache line is
  // re-requested in the NEXT K tiles instead of six K tiles later (set by the host launcher, convg8.hip; needs KH * KW <= 16, KW <= 7; LAST member: the
  // other kernels' argument offsets stay as they were)
  unsigned long long tap_seq;
};

__host__ __device__ inline void y5_fastdiv_make(unsigned d, unsigned* m, int* s) {
  int l = 0;
  while (l < 32 && (1ull << l) < d) ++l;   // ceil(log2 d)
  *m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  *s = (l < 1 ? l : 1) | ((l > 0 ? l - 1 : 0) << 8);
}
__device__ __forceinline__ unsigned y5_fastdiv(unsigned n, unsigned m, int s) {
#ifdef Y5_EMU
  const unsigned t = (unsigned)(((unsigned long long)m * n) >> 32);
#else
  const unsigned t = __umulhi(m, n);
#endif
  return (t + ((n - t) >> (s & 0xff))) >> (s >> 8);
}
inline void y5_conv_set_fastdiv(Y5ConvParams& p) {
  y5_fastdiv_make((unsigned)(p.OH * p.OW), &p.dv_ohw_m, &p.dv_ohw_s);
  y5_fastdiv_make((unsigned)p.OW, &p.dv_ow_m, &p.dv_ow_s);
}

#define Y5_CONV_MAXTAB 4096   // max k-pieces in TABLE mode (LDS: 8 B each)

template <typename T, int RB> struct Y5ConvGeom {
  static constexpr int EPP = 16 / (int)sizeof(T);        // elements per 16-byte piece
  static constexpr int BK = RB / (int)sizeof(T);         // K elements per stage
  static constexpr int NSLOT = RB / 16;                  // 16-byte slots per LDS row
  static constexpr int ROWS_PER_INSTR = 1024 / RB;       // rows covered by one 64-lane LDS-DMA instruction
  static constexpr int SCR_ROWB = 32 * (int)sizeof(T) + 16;  // epilogue scratch row: 32 channels + 16 B pad
  static constexpr int SCR_BYTES = 32 * SCR_ROWB;        // per wave
  __device__ static __forceinline__ int swz(int row) { return RB == 64 ? ((row >> 2) & 3) : ((row >> 1) & 7); }
};

template <typename T, int WM, int WN, int TM, int TN, int RB, int NS = 2, bool ALIAS = false>
constexpr size_t y5_conv_lds_bytes(int table_pieces) {
  return NS * (size_t)(WM * TM * 32 + WN * TN * 32) * RB + (size_t)table_pieces * 8 +
         (ALIAS ? 0 : (size_t)WM * WN * Y5ConvGeom<T, RB>::SCR_BYTES) + (NS > 2 ? (size_t)WM * WN * 1024 : 0);
}

// minimum waves per SIMD requested from the register allocator (keeps the accumulators in the unified VGPR file
// and the allocation under the occupancy steps of MI355X_MICROARCH.md "Register files")
constexpr int y5_conv_min_waves(int tm, int tn) { return tm * tn <= 1 ? 5 : tm * tn <= 2 ? 4 : tm * tn <= 4 ? 3 : 2; }

// NS = number of LDS stages.  NS == 2: issue chunk i+1, compute chunk i, drain (vmcnt(0)) + barrier.  NS >= 3: a ring
// with NS-1 chunks in flight, retired by COUNTED vmcnt (every wave issues the same number of LDS-DMA instructions per
// chunk: filter instructions a wave does not own go to a 1 KiB per-wave dummy slot) and ONE raw s_barrier per chunk.
//
// PROD: producer / consumer specialisation.  One LDS-DMA instruction blocks its wave for 60-180 cycles at issue (the
// vector-memory path takes 64 B/clk/CU), which in the plain kernel is time the wave's SIMD spends without MFMAs.  With
// PROD the workgroup has 2 * WM * WN waves: the first WM * WN (consumers) only read fragments, multiply and run the
// epilogue; the others (producers, one per SIMD beside a consumer) only issue the ring's LDS-DMA instructions, NS-1
// chunks ahead, and retire them with counted vmcnt.  One raw s_barrier per chunk joins the two roles.
//
// ALIAS (2-stage only): the epilogue's transposition scratch lives in the ring stage that is idle at a tile boundary instead
// of in LDS of its own, and one more wave per SIMD is requested from the register allocator -- more workgroups per CU, so
// that layers whose tile count is just above the resident-workgroup count finish in one round instead of two.  The
// prefetch of the new tile's second chunk starts after the epilogue (one extra barrier per tile).
//
// SK (2-stage only): stream-K decomposition.  A layer of yolov5s at bs = 64 has 200 .. 1600 output tiles for 256 .. 512 resident
// workgroups, so the last round of a tile-per-workgroup schedule is 22-56 % empty (M = 25600 = 2^10 * 25 pixels at P5: no power-of-two
// tile shape divides the work evenly).  With SK the unit of work is one K chunk of one tile: workgroup g owns the contiguous range
// [g * U / G, (g + 1) * U / G) of the U = tiles * nk units in (tile, k) order, so every workgroup multiplies the same number of chunks.
// A range boundary inside a tile splits that tile's K loop between consecutive workgroups: the workgroup holding the FIRST k-part
// (always the last thing it does) is the tile's owner; every later part is the FIRST thing its workgroup does -- it parks its fp32
// partial sums in its slab of `sk_ws` (write-through stores), raises its flag and moves on.  The owner, having finished its own
// part long after, polls each flag (one lane, relaxed agent-scope load, workgroup barrier), adds the slabs with cache-bypassing loads,
// clears the flags for the next launch and runs the ordinary epilogue.  Dependencies only point
// from a workgroup to higher-numbered ones that started at the same time (or are dispatched as lower-numbered ones retire), so there
// is nothing to deadlock on; the poll is bounded all the same.
template <typename T, int WM, int WN, int TM, int TN, int RB, bool TABLE, int NS = 2, bool PROD = false, bool ALIAS = false, bool SK = false, bool UP2 = false>
__global__ __launch_bounds__(WM * WN * 64 * (PROD ? 2 : 1), y5_conv_min_waves(TM, TN) + (ALIAS ? 1 : 0))
void y5_conv_igemm_kernel(const Y5ConvParams p) {
  static_assert(!UP2 || (!TABLE && !SK && sizeof(T) == 2), "the virtual upsample + concat loader is built for the uniform fp16 loader");
  static_assert(!PROD || NS >= 3, "producer/consumer needs a ring");
  static_assert(!SK || (NS == 2 && !PROD && !ALIAS), "stream-K is implemented for the plain 2-stage kernel");
  static_assert(!ALIAS || (NS == 2 && !PROD), "scratch aliasing is implemented for the 2-stage kernel");
  static_assert(!ALIAS || (WM * TM * 32 + WN * TN * 32) * RB >= WM * WN * Y5ConvGeom<T, RB>::SCR_BYTES, "stage too small for the scratch");
  using Gm = Y5ConvGeom<T, RB>;
  constexpr int EPP = Gm::EPP, BK = Gm::BK, NSLOT = Gm::NSLOT, RPI = Gm::ROWS_PER_INSTR;
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int ACT_INSTR = BM / RPI, WGT_INSTR = BN / RPI;
  static_assert(ACT_INSTR % NW == 0, "activation tile must split evenly over the waves");
  static_assert(sizeof(T) == 2 || RB == 64, "fp32 path uses 64-byte rows only");
  constexpr int ACT_PER_WAVE = ACT_INSTR / NW;
  constexpr int WGT_PER_WAVE = (WGT_INSTR + NW - 1) / NW;
  constexpr int BUF_BYTES = (BM + BN) * RB;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [buf0: act BM rows | wgt BN rows][buf1: same]..[tap table (TABLE mode)][per-wave epilogue scratch][dummy]
  int2* tab = reinterpret_cast<int2*>(smem + NS * BUF_BYTES);
  const int tab_bytes = TABLE ? (p.Kpad / EPP) * 8 : 0;

#ifdef Y5_DBG_TIMING
  const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = PROD && wave_id >= NW;
  const int wave = producer ? wave_id - NW : wave_id;  // index inside the role: loader share / consumer sub-tile
  const int wm = wave / WN, wn = wave % WN;
  char* scratch = smem + (ALIAS ? 0 : NS * BUF_BYTES + tab_bytes) + wave * Gm::SCR_BYTES;  // ALIAS: re-pointed at every tile boundary
  char* dummy = smem + NS * BUF_BYTES + tab_bytes + NW * Gm::SCR_BYTES + wave * 1024;  // NS > 2 only

  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int ntiles = p.tilesM * p.tilesN;
  const int nk = p.nk;
  // SK: this workgroup's unit range -> first tile, first chunk inside it, number of tiles touched, chunk where the last one stops
  long long sk_u0 = 0, sk_u1 = 0;
  int sk_tf = 0, sk_kb0 = 0, sk_ke_last = nk;
  if constexpr (SK) {
    const long long U = (long long)ntiles * nk;
    sk_u0 = U * bid / G;
    sk_u1 = U * (bid + 1) / G;
    sk_tf = (int)(sk_u0 / nk);
    sk_kb0 = (int)(sk_u0 - (long long)sk_tf * nk);
    sk_ke_last = (int)((sk_u1 - 1) % nk) + 1;
  }
  const int nmine = SK ? (sk_u1 > sk_u0 ? (int)((sk_u1 - 1) / nk) - sk_tf + 1 : 0) : (ntiles - bid + G - 1) / G;  // host guarantees G <= ntiles
  const int total = SK ? (int)(sk_u1 - sk_u0) : nmine * nk;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t wrs = y5_make_rsrc(p.w, p.w_bytes);
  const y5_rsrc_t xrs2 = UP2 ? y5_make_rsrc(p.x2, p.x2_bytes) : xrs;
  constexpr int ES = (int)sizeof(T);

  if constexpr (TABLE) {
    // per k-piece (EPP elements) gather table: {element offset relative to pixel (ih0,iw0), kh | kw<<16}
    const int npieces = p.Kpad / EPP;
    for (int q = tid; q < npieces; q += NW * 64 * (PROD ? 2 : 1)) {
      const int k = q * EPP;
      int2 e;
      if (k < p.K) {
        const int tap = k / p.C1, c = k - tap * p.C1;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        e.x = (kh * p.W + kw) * p.ldx + c;
        e.y = kh | (kw << 16);
      } else {
        e.x = 0;
        e.y = 0x7fff;  // kh = 32767 -> never inside the image
      }
      tab[q] = e;
    }
  }

  // ---- stage-side (loader) state: describes the tile whose K chunks are currently being staged ------------
  const int lrow = lane / NSLOT;   // row inside one LDS-DMA instruction
  const int lslot = lane % NSLOT;  // destination 16-byte slot
  int a_base[ACT_PER_WAVE];        // BYTE offset of (b, ih0, iw0) + source slot (may be negative for padded origins)
  int a_base2[UP2 ? ACT_PER_WAVE : 1];  // UP2: byte offset of the low-resolution pixel (b, oh >> 1, ow >> 1) in x2 + source slot
  int a_ih0[ACT_PER_WAVE], a_iw0[ACT_PER_WAVE], a_slot[ACT_PER_WAVE];  // TABLE mode
  unsigned a_mask[ACT_PER_WAVE];   // UNIFORM mode: bit (kh*KW + kw) CLEAR <=> that tap of this row lies inside the image
  unsigned w_off[WGT_PER_WAVE];    // byte offset of the filter row + source slot, Y5_OOB for rows beyond the tile / Npad
  int u_kh = 0, u_kw = 0, u_c0 = 0;  // uniform tap walker (UNIFORM mode: C1 % BK == 0)
  int s_t = 0, s_kc = SK ? sk_kb0 : 0;  // staged tile (index into this block's tile list) / K chunk

  auto tile_coords = [&](int j, int& m0, int& n0) {
    const int t = SK ? sk_tf + j : y5_xcd_remap(bid + j * G, ntiles);
    const int tn = t % p.tilesN, tm = t / p.tilesN;
    m0 = tm * BM;
    n0 = tn * BN;
  };

  auto loader_setup = [&](int j) __attribute__((always_inline)) {   // (outlined as a real call -- 600 B of stack per wave -- in the 256-row kernels otherwise)
    int m0, n0;
    tile_coords(j, m0, n0);
#pragma unroll
    for (int i = 0; i < ACT_PER_WAVE; ++i) {
      const int row = (wave + i * NW) * RPI + lrow;  // row inside the activation tile
      const int sslot = lslot ^ Gm::swz(row);
      const int m = m0 + row;
      const int mm = m < p.M ? m : 0;
      const int ohw = p.OH * p.OW;
      const int b = (int)y5_fastdiv((unsigned)mm, p.dv_ohw_m, p.dv_ohw_s);
      const int r = mm - b * ohw;
      const int oh = (int)y5_fastdiv((unsigned)r, p.dv_ow_m, p.dv_ow_s), ow = r - oh * p.OW;
      const int ih0 = oh * p.SH - p.PH, iw0 = ow * p.SW - p.PW;
      a_base[i] = (((b * p.H + ih0) * p.W + iw0) * p.ldx + sslot * EPP) * ES;
      if constexpr (UP2) a_base2[i] = (((b * (p.H >> 1) + (oh >> 1)) * (p.W >> 1) + (ow >> 1)) * p.ldx2 + sslot * EPP) * ES;
      if constexpr (TABLE) {
        a_ih0[i] = m < p.M ? ih0 : -0x40000000;  // pixel rows past M: every tap reads zeros
        a_iw0[i] = iw0;
        a_slot[i] = sslot;
      } else {
        // taps inside the image: kh in [max(0, -ih0), min(KH, H - ih0)), kw likewise -- two bit ranges and one shift-or per filter ROW instead
        // of two compares per TAP (round 4: this runs once per tile in the waves that feed the ring; a 1x1 layer's tile is only 8 chunks long)
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
        a_mask[i] = ~mk;  // stored inverted: bit set <=> tap outside the image
      }
    }
#pragma unroll
    for (int i = 0; i < WGT_PER_WAVE; ++i) {
      const int row = (wave + i * NW) * RPI + lrow;  // row inside the filter tile
      const int sslot = lslot ^ Gm::swz(row);
      const int n = n0 + row;
      // invalid rows carry bit 31: every offset >= 2^31 is past the buffer's range (tensors are < 2^31 bytes) and reads as zero, so
      // the per-chunk K offset can be added without a test
      w_off[i] = (row < BN) && (n < p.Npad) ? (unsigned)((n * p.Kpad + sslot * EPP) * ES) : 0x80000000u;
    }
    u_kh = 0; u_kw = 0; u_c0 = 0;
    if constexpr (SK && !TABLE) {
      if (j == 0 && sk_kb0 > 0) {  // the range starts inside this tile: put the uniform tap walker on chunk sk_kb0
        const int k = sk_kb0 * BK;
        const int tap = k / p.C1;
        u_c0 = k - tap * p.C1;
        u_kh = tap / p.KW;
        u_kw = tap - u_kh * p.KW;
      }
    }
  };

  // One chunk = LPC LDS-DMA instructions per wave.  The vector-memory path moves 64 B/clk/CU, so a chunk's loads occupy it
  // for about as long as its MFMAs occupy the matrix cores: stage() issues them back to back (used ahead of an epilogue),
  // stage_begin/stage_load/stage_end let compute() spread them between the MFMAs of the chunk being multiplied.
  constexpr int LPC = ACT_PER_WAVE + WGT_PER_WAVE;  // LDS-DMA instructions per chunk per wave (NS > 2: uniform)
  char* st_lds = smem;
  int st_tap_off = 0, st_tap_bit = 0;
  bool st_up = false;
  unsigned st_kcb = 0;
  auto stage_begin = [&](int buf) __attribute__((always_inline)) {
    st_lds = smem + buf * BUF_BYTES;
    if constexpr (!TABLE) {
      st_tap_off = ((u_kh * p.W + u_kw) * p.ldx + u_c0) * ES;
      st_tap_bit = u_kh * p.KW + u_kw;
      if constexpr (UP2) st_up = u_c0 < p.up_c;
    }
    st_kcb = (unsigned)(s_kc * BK * ES);
  };
  auto stage_load = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q < ACT_PER_WAVE) {
      constexpr int i = q;
      unsigned voff;
      if constexpr (TABLE) {
        const int2 e = tab[s_kc * NSLOT + a_slot[i]];
        const int ih = a_ih0[i] + (e.y & 0xffff), iw = a_iw0[i] + (e.y >> 16);
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        voff = ok ? (unsigned)(a_base[i] + (e.x - a_slot[i] * EPP) * ES) : Y5_OOB;
      } else {
        // outside-the-image taps get bit 31 (= out of range = zero fill): shift, shift-or, add -- no compare / select
        voff = (unsigned)(a_base[i] + st_tap_off) | ((a_mask[i] >> st_tap_bit) << 31);
      }
      if constexpr (UP2) {
        // 1x1 layer: st_tap_off is the chunk's channel offset in bytes; the first up_c channels come from the low-resolution tensor
        // (wave-uniform choice: a chunk never straddles the boundary, up_c % BK == 0)
        if (st_up) {
          y5_bglds16(xrs2, (unsigned)(a_base2[i] + st_tap_off) | (a_mask[i] << 31), st_lds + (wave + i * NW) * 1024);
          return;
        }
      }
      y5_bglds16(xrs, voff, st_lds + (wave + i * NW) * 1024);
    } else {
      constexpr int i = q - ACT_PER_WAVE;
      const int idx = wave + i * NW;
      if (WGT_INSTR % NW == 0 || idx < WGT_INSTR) {
        y5_bglds16(wrs, w_off[i] + st_kcb, st_lds + BM * RB + idx * 1024);
      } else if (NS > 2) {
        y5_bglds16_dummy(wrs, dummy);  // keeps the per-chunk LDS-DMA count identical in every wave (counted vmcnt)
      }
    }
  };
  auto stage_end = [&]() __attribute__((always_inline)) {  // advance to the next (tile, chunk)
    if constexpr (!TABLE) {
      u_c0 += BK;
      if (u_c0 >= p.C1) {
        u_c0 = 0;
        if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
      }
    }
    if (++s_kc == nk) {
      s_kc = 0;
      if (++s_t < nmine) loader_setup(s_t);
    }
  };
  auto stage = [&](int buf) __attribute__((always_inline)) {
#ifdef Y5_DBG_NOLOAD
    if (s_t > 0 || s_kc > 0) { stage_end(); return; }
#endif
    stage_begin(buf);
    y5_static_for<0, LPC>([&](auto qc) { stage_load(qc); });
    stage_end();
  };

  float16_t acc[TM][TN];

  // fragment read offsets (bytes inside a buffer); lane -> row (lane&31), k-slot group g = lane>>5
  const int g = lane >> 5;
  const int frow = lane & 31;
  int a_rd[TM], w_rd[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_rd[i] = (wm * TM * 32 + i * 32 + frow) * RB;
#pragma unroll
  for (int j = 0; j < TN; ++j) w_rd[j] = BM * RB + (wn * TN * 32 + j * 32 + frow) * RB;
  const int fsw = Gm::swz(frow);  // swizzle term of this lane's rows (tile bases are multiples of 32)

  // ---- epilogue of one finished tile: bias + act -> per-wave LDS transpose -> 16-byte coalesced stores -----
  T* yg = static_cast<T*>(p.y);  // may alias p.res (in-place residual): no restrict
  const T* rg = static_cast<const T*>(p.res);
  T* y2g = static_cast<T*>(p.y2);
  constexpr int SCR_ROWB = Gm::SCR_ROWB;
  constexpr int CPV = 16 / (int)sizeof(T);           // channels per 16-byte vector
  constexpr int VPR = 32 / CPV;                      // vectors per scratch row (4 for half, 8 for float)
  constexpr int ROWS_PER_PASS = 64 / VPR;            // 16 (half) / 8 (float)
  auto epilogue = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mt = m0 + wm * TM * 32 + i * 32;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nt = n0 + wn * TN * 32 + j * 32;
        // (1) registers -> scratch[pixel row = lane&31][channel]
        // bias through the SCALAR cache (nt is wave-uniform; eight consecutive values per s_load, the lane picks its half): a
        // vector load here would sit behind the next tile's in-flight LDS-DMA in the in-order vmcnt queue and stall the epilogue
        // for a full memory round trip
        const float* pb = p.bias + (nt < p.Npad ? nt : 0);
        const float keep = nt < p.Npad ? 1.0f : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = q * 8 + g * 4;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = acc[i][j][q * 4 + e] + (g ? pb[q * 8 + 4 + e] : pb[q * 8 + e]) * keep;
            v[e] = p.act ? y5_silu(t) : t;
          }
          char* dst = scratch + frow * SCR_ROWB + nl * (int)sizeof(T);
          if constexpr (sizeof(T) == 2) {
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
            *reinterpret_cast<half4_t*>(dst) = o;
          } else {
            float4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e];
            *reinterpret_cast<float4_t*>(dst) = o;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (2) scratch rows -> global, 16 B per lane, VPR lanes cover one pixel's 32 channels
#pragma unroll
        for (int ps = 0; ps < 32 / ROWS_PER_PASS; ++ps) {
          const int row = ps * ROWS_PER_PASS + lane / VPR;
          const int vs = lane % VPR;
          const int m = mt + row;
          const int n = nt + vs * CPV;
          if (m < p.M && n < p.C2) {
            uint4_t raw = *reinterpret_cast<const uint4_t*>(scratch + row * SCR_ROWB + vs * 16);
            size_t mo = (size_t)m;
            if (p.o_mul_h) {
              const int ohw = p.OH * p.OW;
              const int b = m / ohw;
              const int r = m - b * ohw;
              const int oh = r / p.OW, ow = r - oh * p.OW;
              mo = ((size_t)b * p.o_H + oh * p.o_mul_h + p.o_off_h) * p.o_W + ow * p.o_mul_w + p.o_off_w;
            }
            if (rg) {
              const uint4_t rr = *reinterpret_cast<const uint4_t*>(rg + mo * p.ldr + n);
              if constexpr (sizeof(T) == 2) {
                half8_t a = __builtin_bit_cast(half8_t, raw), b = __builtin_bit_cast(half8_t, rr), c;
#pragma unroll
                for (int e = 0; e < 8; ++e) c[e] = (half_t)((float)a[e] + (float)b[e]);
                raw = __builtin_bit_cast(uint4_t, c);
              } else {
                float4_t a = __builtin_bit_cast(float4_t, raw), b = __builtin_bit_cast(float4_t, rr);
                a += b;
                raw = __builtin_bit_cast(uint4_t, a);
              }
            }
            if (p.split_n) {
              T* d = n < p.split_n ? yg + mo * p.ldy + n : y2g + mo * p.ld2 + (n - p.split_n);
              *reinterpret_cast<uint4_t*>(d) = raw;
            } else {
            if (yg) *reinterpret_cast<uint4_t*>(yg + mo * p.ldy + n) = raw;
            if (y2g) {
              const int ohw = p.OH * p.OW;
              const int b = m / ohw;
              const int r = m - b * ohw;
              const int oh = r / p.OW, ow = r - oh * p.OW;
              const size_t row0 = ((size_t)b * 2 * p.OH + 2 * oh) * (2 * p.OW) + 2 * ow;
              T* d0 = y2g + row0 * p.ld2 + n;
              T* d1 = y2g + (row0 + 2 * p.OW) * p.ld2 + n;
              *reinterpret_cast<uint4_t*>(d0) = raw;
              *reinterpret_cast<uint4_t*>(d0 + p.ld2) = raw;
              *reinterpret_cast<uint4_t*>(d1) = raw;
              *reinterpret_cast<uint4_t*>(d1 + p.ld2) = raw;
            }
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  };

  if constexpr (TABLE) __syncthreads();
  if (!PROD || producer) loader_setup(0);
  // multiply the chunk in `lds`; with_loads: the next chunk's LDS-DMA instructions (stage_begin() already called) are issued
  // between the k-steps, so that the vector-memory pipe and the matrix cores work at the same time
  auto compute = [&](const char* lds, bool with_loads) {
#ifdef Y5_DBG_NOMFMA
    asm volatile("" :: "v"(lds));
    if (with_loads) y5_static_for<0, LPC>([&](auto qc) { stage_load(qc); });
    return;
#endif
    if constexpr (sizeof(T) == 2) {
      // Software-pipelined over the k-steps of the chunk: the fragments of k-step ks+1 are fetched from LDS BEFORE the MFMAs of k-step ks
      // are issued (second register set), and the next chunk's LDS-DMA pieces are issued AFTER them -- the matrix cores work through the
      // LDS round trip and through the 100+ cycles an LDS-DMA instruction holds its wave at issue.  (Left to the compiler the loop was
      // read -> DMA -> wait -> 4 MFMAs per k-step with one register set: the LDS-DMA builtins are LDS writes it will not move reads across.)
      constexpr int KS = RB / 32, LPK = (LPC + KS - 1) / KS;
      half8_t af[2][TM], wf[2][TN];
      auto rd = [&](auto ksc, auto bc) {
        constexpr int ks = decltype(ksc)::value, b = decltype(bc)::value;
        const int so = ((ks * 2 + g) ^ fsw) * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[b][i] = *reinterpret_cast<const half8_t*>(lds + a_rd[i] + so);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[b][j] = *reinterpret_cast<const half8_t*>(lds + w_rd[j] + so);
      };
      rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      y5_static_for<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value, b = ks & 1;
        if constexpr (ks + 1 < KS) rd(std::integral_constant<int, ks + 1>{}, std::integral_constant<int, b ^ 1>{});
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[b][j], af[b][i], acc[i][j], 0, 0, 0);
#ifndef Y5_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (with_loads) {
          constexpr int Q0 = ks * LPK < LPC ? ks * LPK : LPC, Q1 = (ks + 1) * LPK < LPC ? (ks + 1) * LPK : LPC;
          y5_static_for<Q0, Q1>([&](auto qc) { stage_load(qc); });
        }
      });
    } else {
      if (with_loads) y5_static_for<0, LPC>([&](auto qc) { stage_load(qc); });
      const int so0 = ((2 * g) ^ fsw) * 16, so1 = ((2 * g + 1) ^ fsw) * 16;
      float4_t af[TM][2], wf[TN][2];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        af[i][0] = *reinterpret_cast<const float4_t*>(lds + a_rd[i] + so0);
        af[i][1] = *reinterpret_cast<const float4_t*>(lds + a_rd[i] + so1);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        wf[j][0] = *reinterpret_cast<const float4_t*>(lds + w_rd[j] + so0);
        wf[j][1] = *reinterpret_cast<const float4_t*>(lds + w_rd[j] + so1);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][h][e], af[i][h][e], acc[i][j], 0, 0, 0);
    }
  };
  auto tile_begin = [&](int ti, int& pm0, int& pn0) {
    if (ti > 0) epilogue(pm0, pn0);
    tile_coords(ti, pm0, pn0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  int pm0 = 0, pn0 = 0;  // coordinates of the tile whose results sit in acc
#ifdef Y5_DBG_TIMING
  unsigned long long r_start = 0, r_loop_end = 0;
#endif
  if constexpr (PROD) {
    if (producer) {
      for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < total) stage(s0);
      int nxt = NS - 1;
      for (int it = 0; it < total; ++it) {
        const int ahead = total - 1 - it < NS - 2 ? total - 1 - it : NS - 2;  // chunks issued after chunk `it`
        switch (ahead) {
          case 0: y5_wait_vm<0>(); break;
          case 1: y5_wait_vm<LPC>(); break;
          case 2: y5_wait_vm<2 * LPC>(); break;
          default: y5_wait_vm<3 * LPC>(); break;
        }
        __builtin_amdgcn_s_barrier();  // chunk `it` is in LDS for the consumers; they are done with chunk it-1 (slot nxt)
        if (it + NS - 1 < total) stage(nxt);
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
      }
      return;
    }
    // Consumer loop, software-pipelined ACROSS the chunk boundary (round 4).  The barrier that says "chunk c+1 has landed" is taken before the
    // LAST k-step of chunk c instead of after it, and that k-step's MFMAs run behind the fragment reads of (c+1, k-step 0): a chunk no longer
    // opens with an LDS round trip during which the wave's matrix-core slot idles (8 reads, ~130 cycles, then 8 MFMAs = 256 cycles per BK32
    // chunk before).  The ring makes it legal: the producers run NS-1 chunks ahead, so chunk c+1 is normally in LDS long before it is asked for.
    // Hand-over of the slot of chunk c to the producers happens at that same barrier, so every read of chunk c must have RETURNED by then
    // (lgkmcnt(0): the last reads were issued a k-step = 4+ MFMAs earlier).  Same number of barriers in the same order as before: the
    // producer side is unchanged.  At a tile boundary the first chunk is read after the barrier as before (the epilogue sits in between).
    constexpr int KS = RB / 32;
    static_assert(KS % 2 == 0, "two fragment register sets alternate per k-step: a chunk must hold an even number of k-steps");
    int cur = 0;
    half8_t af[2][TM], wf[2][TN];
    auto rd = [&](const char* lds, auto ksc, auto bc) {
      constexpr int ks = decltype(ksc)::value, b = decltype(bc)::value;
      const int so = ((ks * 2 + g) ^ fsw) * 16;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[b][i] = *reinterpret_cast<const half8_t*>(lds + a_rd[i] + so);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[b][j] = *reinterpret_cast<const half8_t*>(lds + w_rd[j] + so);
    };
    for (int ti = 0; ti < nmine; ++ti) {
      __builtin_amdgcn_s_barrier();
      tile_begin(ti, pm0, pn0);
      rd(smem + cur * BUF_BYTES, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      for (int kc = 0; kc < nk; ++kc) {
        const char* lds = smem + cur * BUF_BYTES;
        const int nxt_buf = cur + 1 == NS ? 0 : cur + 1;
        y5_static_for<0, KS>([&](auto ksc) {
          constexpr int ks = decltype(ksc)::value, b = ks & 1;
          if constexpr (ks + 1 < KS) {
            rd(lds, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, b ^ 1>{});
          } else {
            if (kc + 1 < nk) {
              __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): every fragment of chunk kc is in registers before its slot is handed back
              __builtin_amdgcn_s_barrier();        // chunk kc + 1 has landed
              rd(smem + nxt_buf * BUF_BYTES, std::integral_constant<int, 0>{}, std::integral_constant<int, b ^ 1>{});
            }
          }
#ifndef Y5_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[b][j], af[b][i], acc[i][j], 0, 0, 0);
#ifndef Y5_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
        });
        cur = nxt_buf;
      }
    }
  } else if constexpr (SK) {
    // ---- stream-K main loop: `total` chunk-units in (tile, k) order, first tile entered at chunk sk_kb0 --------------------
    // Slabs travel with sc1 (write-through / cache-bypassing) 16-byte buffer stores and loads: no agent-scope fence anywhere (a
    // release fence is a write-back of the XCD's whole L2 -- every workgroup doing one made the first version 2x slower than no
    // stream-K at all).  Publish = stores, s_waitcnt vmcnt(0) in every wave, workgroup barrier, relaxed agent-scope flag store;
    // consume = relaxed agent-scope poll by one lane, workgroup barrier, sc1 loads (cdna guide, split-K recipe, second form).
    constexpr int SLAB_BYTES = NW * TM * TN * 16 * 64 * 4;  // per workgroup
    constexpr int SC1 = 16;
    auto slab_rsrc = [&](int g) { return y5_make_rsrc(reinterpret_cast<const char*>(p.sk_ws) + (size_t)g * SLAB_BYTES, (unsigned)SLAB_BYTES); };
    auto park_partial = [&]() {  // this workgroup holds a LATER k-part of the tile: publish the sums, the owner adds them
      const y5_rsrc_t rs = slab_rsrc(bid);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[i][j][q * 4 + e]);
            y5_buffer_store16(v, rs, ((((wave * TM + i) * TN + j) * 4 + q) * 64 + lane) * 16, SC1);
          }
      Y5_DRAIN_VM();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.sk_flags + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto gather_partials = [&](int tile) {  // owner: add the slabs of the workgroups whose ranges start inside `tile`
      const long long tile_end = (long long)(tile + 1) * nk, U = (long long)ntiles * nk;
      for (int g2 = bid + 1; g2 < G && U * g2 / G < tile_end; ++g2) {
        if (tid == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(p.sk_flags + g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        const y5_rsrc_t rs = slab_rsrc(g2);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint4_t v = y5_buffer_load16(rs, ((((wave * TM + i) * TN + j) * 4 + q) * 64 + lane) * 16, SC1);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[i][j][q * 4 + e] += __uint_as_float(v[e]);
            }
        __syncthreads();  // every wave has read the slab before the flag lets its writer (next launch) reuse it
        if (tid == 0) __hip_atomic_store(p.sk_flags + g2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    // kind of the segment whose sums sit in acc: 0 = whole tile, 1 = later k-part (park), 2 = first k-part of a split tile (owner)
    auto finish_segment = [&](int ti) {
      const int kb = ti == 0 ? sk_kb0 : 0, ke = ti == nmine - 1 ? sk_ke_last : nk;
      if (kb > 0) { park_partial(); return; }
      if (ke < nk) gather_partials(sk_tf + ti);
      epilogue(pm0, pn0);
    };
    if (total > 0) {
      stage(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      __syncthreads();
      int ti = 0, kc = sk_kb0;
      for (int it = 0; it < total; ++it) {
        const int cur = it & 1;
        const bool more = it + 1 < total;
        const bool seg_start = it == 0 || kc == 0;
        const bool spread = more && !seg_start;
        if (more) { if (spread) stage_begin(cur ^ 1); else stage(cur ^ 1); }
        if (seg_start) {
          if (it > 0) finish_segment(ti - 1);
          tile_coords(ti, pm0, pn0);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        compute(smem + cur * BUF_BYTES, spread);
        if (spread) stage_end();
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (++kc == nk) { kc = 0; ++ti; }
      }
      finish_segment(nmine - 1);
    }
    return;
  } else if constexpr (NS == 2) {
    stage(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    int it = 0;
#ifdef Y5_DBG_TIMING
    unsigned long long d_stage = 0, d_comp = 0, d_wait = 0, d_bar = 0, d_epi = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    r_start = __builtin_amdgcn_s_memrealtime();
#endif
    for (int ti = 0; ti < nmine; ++ti) {
      for (int kc = 0; kc < nk; ++kc, ++it) {
        const int cur = it & 1;
#ifdef Y5_DBG_TIMING
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // next chunk (possibly of the next tile): ahead of an epilogue all at once (it flies during the epilogue), otherwise
        // spread over this chunk's k-steps
        const bool more = it + 1 < total;
        const bool spread = more && kc != 0;
        if constexpr (ALIAS) {
          if (kc == 0) {
            scratch = smem + (cur ^ 1) * BUF_BYTES + wave * Gm::SCR_BYTES;  // stage cur^1 held the previous chunk: idle
            tile_begin(ti, pm0, pn0);
            if (ti > 0) __syncthreads();  // every wave is done with its scratch before the stage is refilled
          }
        }
        if (more) { if (spread) stage_begin(cur ^ 1); else stage(cur ^ 1); }
#ifdef Y5_DBG_TIMING
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        if (!ALIAS && kc == 0) tile_begin(ti, pm0, pn0);
#ifdef Y5_DBG_TIMING
        const unsigned long long t1b = __builtin_amdgcn_s_memtime();
#endif
        compute(smem + cur * BUF_BYTES, spread);
        if (spread) stage_end();
#ifdef Y5_DBG_TIMING
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the chunk staged above has landed
#ifdef Y5_DBG_TIMING
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef Y5_DBG_TIMING
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        d_stage += t1 - t0; d_epi += t1b - t1; d_comp += t2 - t1b; d_wait += t3 - t2; d_bar += t4 - t3;
#endif
      }
    }
#ifdef Y5_DBG_TIMING
    if (blockIdx.x == 0 && lane == 0) {
      unsigned long long* o = y5_dbg_timing + wave * 8;
      o[0] = d_stage; o[1] = d_epi; o[2] = d_comp; o[3] = d_wait; o[4] = d_bar; o[5] = __builtin_amdgcn_s_memtime() - t_start; o[6] = total; o[7] = nmine;
      y5_dbg_timing[64 + wave] = __builtin_amdgcn_s_memrealtime() - r_start;
    }
    r_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
  } else {
    // ring: chunks it+1 .. it+NS-2 are in flight while chunk `it` is retired; chunk it+NS-1 is issued right after the barrier
    // into the buffer chunk it-1 occupied (all waves have finished reading it: they are past this barrier).
    for (int s0 = 0; s0 < NS - 1; ++s0)
      if (s0 < total) stage(s0);
    int it = 0, cur = 0, nxt = NS - 1;
    for (int ti = 0; ti < nmine; ++ti) {
      for (int kc = 0; kc < nk; ++kc, ++it) {
        const int ahead = total - 1 - it < NS - 2 ? total - 1 - it : NS - 2;  // chunks issued after chunk `it`
        switch (ahead) {
          case 0: y5_wait_vm<0>(); break;
          case 1: y5_wait_vm<LPC>(); break;
          case 2: y5_wait_vm<2 * LPC>(); break;
          default: y5_wait_vm<3 * LPC>(); break;
        }
        __builtin_amdgcn_s_barrier();
        const bool more = it + NS - 1 < total;
        const bool spread = more && kc != 0;
        if (more) { if (spread) stage_begin(nxt); else stage(nxt); }
        if (kc == 0) tile_begin(ti, pm0, pn0);
        compute(smem + cur * BUF_BYTES, spread);
        if (spread) stage_end();
        cur = cur + 1 == NS ? 0 : cur + 1;
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
      }
    }
  }
  if constexpr (ALIAS) scratch = smem + wave * Gm::SCR_BYTES;  // both stages are idle after the last barrier
  epilogue(pm0, pn0);
#ifdef Y5_DBG_TIMING
  if constexpr (NS == 2 && !PROD) {
    if (tid == 0 && blockIdx.x < 1024) {
      unsigned long long* o = y5_dbg_blocks + blockIdx.x * 4;
      o[0] = r_entry; o[1] = r_start; o[2] = r_loop_end; o[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
#endif
}
