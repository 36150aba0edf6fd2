// Weight gradient of the convolution (autograd of models/common.py:74-92 `Conv`, train.py:410 `scaler.scale(loss).backward()`):
//     dW[n][kh][kw][c] = sum over (b, oh, ow) of dz[b, oh, ow, n] * x[b, oh*s + kh - p, ow*s + kw - p, c]
// written into the packed [Npad][Kpad] filter layout of y5_conv2d_fwd (k = (kh, kw, c)), fp32, accumulated with atomics
// (the caller zero-fills dW).  GEMM view: D[n][k] += sum_p dzT[n][p] * X[p][k] with the contraction over pixels, so both
// MFMA operands need 8 consecutive PIXELS of one channel per lane -- a transposed read of the pixel-major NHWC tensors:
// 32-pixel chunks of dz (64 output channels) and of the gathered x columns (64 k) are staged row-major in LDS by
// buffer-addressed LDS-DMA (image borders / tails = zero fill) and the fragments are gathered with 16-bit LDS reads.
// One workgroup = one (64*TNB) x (64*TKB) tile of dW and one slice of the pixel range (split-K over the grid); every wave
// keeps TNB x TKB accumulators.  The chunk loop is a latency problem (one 32-pixel chunk is 8-16 KiB per workgroup against
// a ~2 us HBM round trip), so chunks travel through an S-stage LDS ring: S-1 chunks are in flight per workgroup, retired
// with counted s_waitcnt vmcnt and ONE raw s_barrier per chunk, and the gather coordinates (b, oh, ow) of every staged
// row advance incrementally (no integer division in the loop).
#include <hip/hip_runtime.h>

#include <cstdlib>


#include "../../include/yolov5_hip.h"
#include "y5_common.h"
#include "y5_host.h"

struct Y5WgradParams {
  const void* x; const void* dz; float* dw;
  unsigned x_bytes, dz_bytes;
  int B, H, W, C1, ldx, OH, OW, C2, ldz, KH, KW, SH, SW, PH, PW, K, Kpad, Npad;
  int M;            // B*OH*OW
  int tiles_n, tiles_k, splits;
  int pix_per_split;  // multiple of 32
  float* ws;          // DET kernels: [splits][Npad][Kpad] partial sums, one slab per pixel-range split (plain stores, no atomics)
};

// LIN: pointwise layers (k1 s1 p0: the x pixel of a staged row IS its output pixel) -- both operands advance by a constant byte step per chunk, so staging
// a row costs an add, a compare and a select instead of the coordinate walk + bounds tests of the general gather (33 of yolov5s' 57 layers; the ablation
// builds put staging -- LDS-DMA issue plus its address arithmetic -- at 40 % of this kernel)
// DET: deterministic reduction -- every (tile, split) workgroup writes its partial tile into ITS slab of p.ws with plain stores; y5_wgrad_reduce_kernel adds
// the slabs in split order.  The default (atomics into dW) is order-dependent in the last bits of fp32 (run-to-run, rank-to-rank).
template <int TNB, int TKB, int S, bool LIN, bool DET = false>
__global__ __launch_bounds__(256)
void y5_conv_wgrad_kernel(const Y5WgradParams p) {
  typedef half_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][dz tile 32 x 128 B | x tile 32 x 128 B]
  constexpr int ZROW = 128 * TNB, XROW = 128 * TKB;            // bytes per staged pixel row (64*T channels)
  constexpr int ZT = 32 * ZROW, XT = 32 * XROW, BUF = ZT + XT;
  constexpr int ZI = ZT / 1024, XI = XT / 1024, NI = (ZI + XI) / 4;  // LDS-DMA instructions per chunk: dz, x, per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  const int tk = bid % p.tiles_k;
  const int tn = (bid / p.tiles_k) % p.tiles_n;
  const int sp = bid / (p.tiles_k * p.tiles_n);
  const int n0 = tn * 64 * TNB, k0 = tk * 64 * TKB;
  const int m_begin = sp * p.pix_per_split;
  const int m_end = m_begin + p.pix_per_split < p.M ? m_begin + p.pix_per_split : p.M;
  if (m_begin >= m_end) return;
  const int nchunks = (m_end - m_begin + 31) / 32;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t zrs = y5_make_rsrc(p.dz, p.dz_bytes);

  // staging roles: ZI + XI LDS-DMA instructions per chunk (1 KiB each), NI per wave; a dz instruction covers 1024/ZROW
  // pixel rows of 64*TNB channels, an x instruction 1024/XROW rows of 64*TKB gathered k columns
  const int ohw = p.OH * p.OW;
  const int q32 = 32 / p.OW, r32 = 32 - q32 * p.OW;   // a chunk advances every staged row by 32 pixels = q32 rows + r32 columns
  bool i_isx[NI], i_ok[NI];
  int i_dst[NI], i_m[NI];       // LDS offset; pixel index of the row staged next
  int z_off[NI];                // dz: byte offset of the lane's 16 bytes inside a pixel row
  int x_b[NI], x_oh[NI], x_ow[NI], x_dh[NI], x_dw[NI], x_c[NI];  // x: pixel coordinates, tap offset (kh - PH, kw - PW), channel
  unsigned l_off[NI], l_step[NI];                                // LIN: byte offset of the row staged next, its step per chunk
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int I = wave * NI + q;
    const bool isx = I >= ZI;
    const int J = isx ? I - ZI : I;
    const int rowb = isx ? XROW : ZROW;
    const int lpr = rowb / 16;                       // lanes per pixel row
    const int r = J * (1024 / rowb) + lane / lpr;    // pixel row inside the chunk
    const int s8 = lane % lpr;                       // 16-byte group inside the row
    i_isx[q] = isx; i_dst[q] = (isx ? ZT : 0) + J * 1024; i_m[q] = m_begin + r;
    z_off[q] = (n0 + 8 * s8) * 2;
    x_b[q] = x_oh[q] = x_ow[q] = x_dh[q] = x_dw[q] = x_c[q] = 0;
    if (isx) {
      const int kx = k0 + 8 * s8;
      i_ok[q] = kx < p.K;
      const int tap = i_ok[q] ? kx / p.C1 : 0;
      x_c[q] = kx - tap * p.C1;
      const int kh = tap / p.KW;
      x_dh[q] = kh - p.PH; x_dw[q] = tap - kh * p.KW - p.PW;
      x_b[q] = i_m[q] / ohw;
      const int rr = i_m[q] - x_b[q] * ohw;
      x_oh[q] = rr / p.OW; x_ow[q] = rr - x_oh[q] * p.OW;
    } else {
      i_ok[q] = n0 + 8 * s8 < p.C2;
    }
    if constexpr (LIN) {
      const int ld = isx ? p.ldx : p.ldz;
      l_off[q] = (unsigned)(i_m[q] * ld * 2 + (isx ? x_c[q] * 2 : z_off[q]));
      l_step[q] = (unsigned)(32 * ld * 2);
    }
  }

  // stage the next chunk of this wave's rows into ring slot `buf` and advance the rows by 32 pixels; rows past the end of
  // the split are still issued (out-of-range offset = zero fill) so that every chunk costs exactly NI vmcnt slots
  auto stage = [&](int buf) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      if constexpr (LIN) {
        y5_bglds16(i_isx[q] ? xrs : zrs, (i_m[q] < m_end && i_ok[q]) ? l_off[q] : Y5_OOB, base + i_dst[q]);
        i_m[q] += 32;
        l_off[q] += l_step[q];
        continue;
      }
      unsigned voff = Y5_OOB;
      if (i_m[q] < m_end && i_ok[q]) {
        if (!i_isx[q]) {
          voff = (unsigned)(i_m[q] * p.ldz * 2 + z_off[q]);
        } else {
          const int ih = x_oh[q] * p.SH + x_dh[q], iw = x_ow[q] * p.SW + x_dw[q];
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
            voff = (unsigned)((((x_b[q] * p.H + ih) * p.W + iw) * p.ldx + x_c[q]) * 2);
        }
      }
#ifndef Y5_WG_NOSTAGE
      y5_bglds16(i_isx[q] ? xrs : zrs, voff, base + i_dst[q]);
#else
      if (voff == 12345u) y5_bglds16(i_isx[q] ? xrs : zrs, voff, base + i_dst[q]);
#endif
      i_m[q] += 32;
      if (i_isx[q]) {
        x_ow[q] += r32; x_oh[q] += q32;
        if (x_ow[q] >= p.OW) { x_ow[q] -= p.OW; ++x_oh[q]; }
        while (x_oh[q] >= p.OH) { x_oh[q] -= p.OH; ++x_b[q]; }
      }
    }
  };

  // MFMA roles: wave -> (32*TNB) x (32*TKB) sub-tile (wn, wk) of the block tile, TNB x TKB accumulators
  const int wn = wave >> 1, wk = wave & 1;
  const int fi = lane & 31, g = lane >> 5;
  float16_t acc[TNB][TKB];
#pragma unroll
  for (int a = 0; a < TNB; ++a)
#pragma unroll
    for (int b2 = 0; b2 < TKB; ++b2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.f;

#pragma unroll
  for (int s = 0; s < S - 1; ++s) stage(s);
  int cur = 0, nxt = S - 1;
  for (int ch = 0; ch < nchunks; ++ch) {
    y5_wait_vm<(S - 2) * NI>();            // this wave's share of chunk ch has landed (S-2 younger chunks may still fly)
    __builtin_amdgcn_s_barrier();          // ... and everybody's; all waves are done reading ring slot nxt (chunk ch-1)
    stage(nxt);                            // chunk ch + S - 1
    const T* zt = reinterpret_cast<const T*>(smem + cur * BUF) + wn * 32 * TNB + fi;        // [p][64*TNB]: column n
    const T* xt = reinterpret_cast<const T*>(smem + cur * BUF + ZT) + wk * 32 * TKB + fi;   // [p][64*TKB]: column k
    nxt = cur;
    cur = cur + 1 == S ? 0 : cur + 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8_t af[TNB], bf[TKB];
#ifndef Y5_WG_NOREAD
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int pr = ks * 16 + g * 8 + e;
#pragma unroll
        for (int a = 0; a < TNB; ++a) af[a][e] = zt[pr * (64 * TNB) + a * 32];
#pragma unroll
        for (int b2 = 0; b2 < TKB; ++b2) bf[b2][e] = xt[pr * (64 * TKB) + b2 * 32];
      }
#else
#pragma unroll
      for (int a = 0; a < TNB; ++a) af[a] = __builtin_bit_cast(half8_t, uint4_t{(uint32_t)ch, (uint32_t)lane, 3u, (uint32_t)a});
#pragma unroll
      for (int b2 = 0; b2 < TKB; ++b2) bf[b2] = __builtin_bit_cast(half8_t, uint4_t{(uint32_t)lane, (uint32_t)ch, 5u, (uint32_t)b2});
#endif
#ifndef Y5_WG_NOMFMA
#pragma unroll
      for (int a = 0; a < TNB; ++a)
#pragma unroll
        for (int b2 = 0; b2 < TKB; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b2], acc[a][b2], 0, 0, 0);
#else
#pragma unroll
      for (int a = 0; a < TNB; ++a)
#pragma unroll
        for (int b2 = 0; b2 < TKB; ++b2)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[a][b2][e] += (float)af[a][e] * (float)bf[b2][e];
#endif
    }
  }
  y5_wait_vm<0>();   // drain the zero-fill prefetches past the end of the split before the wave retires its LDS
  // D[i][j]: col j = lane & 31 (k), row i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (n)
#pragma unroll
  for (int b2 = 0; b2 < TKB; ++b2) {
    const int kcol = k0 + wk * 32 * TKB + b2 * 32 + fi;
    if (kcol >= p.K) continue;
#pragma unroll
    for (int a = 0; a < TNB; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 * TNB + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
#ifndef Y5_WG_NOATOM
        if constexpr (DET) {
          if (n < p.C2) p.ws[((size_t)sp * p.Npad + n) * p.Kpad + kcol] = acc[a][b2][r];
        } else if (n < p.C2) atomicAdd(p.dw + (size_t)n * p.Kpad + kcol, acc[a][b2][r]);
#else
        if (n < p.C2 && acc[a][b2][r] == 1.2345f) p.dw[(size_t)n * p.Kpad + kcol] = acc[a][b2][r];
#endif
      }
  }
}

// dW[n][k] += sum over the splits, in split order, of the slabs the DET kernel wrote (fixed summation order: bit-identical run to run)
__global__ void y5_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int C2, int K, int Npad, int Kpad, int splits) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)C2 * Kpad) return;
  const int k = (int)(i % Kpad), n = (int)(i / Kpad);
  if (k >= K) return;
  float s = 0.f;
  for (int sp = 0; sp < splits; ++sp) s += ws[((size_t)sp * Npad + n) * Kpad + k];
  dw[(size_t)n * Kpad + k] += s;
}

#include "wgrad3.h"

// fp32 weight gradient (TrainEngine's reference-precision mode): one thread per packed filter element, a plain fp32 sum over all output
// pixels in a fixed order -- deterministic, exact fp32; not a hot path
__global__ void y5_conv_wgrad_f32_kernel(const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ dw, int B, int H, int W, int C1,
                                         int ldx, int OH, int OW, int C2, int ldz, int KH, int KW, int SH, int SW, int PH, int PW, int Kpad, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % Kpad), n = (int)(i / Kpad);
  if (n >= C2 || k >= KH * KW * C1) return;
  const int c = k % C1, t = k / C1, kh = t / KW, kw = t - kh * KW;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (int oh = 0; oh < OH; ++oh) {
      const int ih = oh * SH - PH + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int ow = 0; ow < OW; ++ow) {
        const int iw = ow * SW - PW + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        s += dz[(((size_t)b * OH + oh) * OW + ow) * ldz + n] * x[(((size_t)b * H + ih) * W + iw) * ldx + c];
      }
    }
  dw[(size_t)n * Kpad + k] += s;
}

template <int NT, int CT, int STR, int S, bool DET>
static void launch_wgrad3(const Y5WgradParams& p, unsigned grid, hipStream_t st) {
  constexpr int XI = ((2 * 3 * (15 * STR + 3)) * 64 * CT + 1023) / 1024;
  constexpr size_t bytes = (size_t)S * (32 * 64 * NT + XI * 1024);
  auto kern = y5_conv_wgrad3_kernel<NT, CT, STR, S, DET>;
  static bool attr_done = false;
  if (!attr_done) {
    if (bytes > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(576), bytes, st, p);
}

static int wgrad_impl(const y5_conv_desc* d, const void* x, const void* dz, int ld_dz, float* dw_packed, float* ws, size_t ws_bytes, size_t* need, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (need) *need = 0;
  if (!d || (!need && (!x || !dz || !dw_packed))) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: null pointer");
  if (d->dtype == Y5_F32) {
    if (need) return Y5_OK;  // the fp32 kernel is a fixed-order sum already: no workspace
    const int oh32 = (d->H + 2 * d->PH - d->KH) / d->SH + 1, ow32 = (d->W + 2 * d->PW - d->KW) / d->SW + 1;
    if (oh32 != d->OH || ow32 != d->OW || d->Kpad < d->KH * d->KW * d->C1 || d->Npad < d->C2) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: inconsistent geometry");
    const long long total = (long long)d->C2 * d->Kpad;
    hipLaunchKernelGGL(y5_conv_wgrad_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)x, (const float*)dz, dw_packed, d->B, d->H,
                       d->W, d->C1, d->ldx, oh32, ow32, d->C2, ld_dz, d->KH, d->KW, d->SH, d->SW, d->PH, d->PW, d->Kpad, total);
    return y5_check_launch("y5_conv2d_wgrad(f32)");
  }
  if (d->dtype != Y5_F16) return y5_fail(Y5_ERR_UNSUPPORTED, "wgrad: fp16 or fp32 activations / gradients only");
  if (d->C1 % 8 || d->ldx % 8 || ld_dz % 8 || d->C2 % 8) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: C1, C2, ldx, ld_dz must be multiples of 8");
  const int oh = (d->H + 2 * d->PH - d->KH) / d->SH + 1, ow = (d->W + 2 * d->PW - d->KW) / d->SW + 1;
  if (oh != d->OH || ow != d->OW) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: OH/OW inconsistent with H/W/k/s/p");
  if (!need && (((uintptr_t)x | (uintptr_t)dz | (uintptr_t)dw_packed) & 15)) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: pointers must be 16-byte aligned");
  const long long xb = (((long long)d->B * d->H * d->W - 1) * d->ldx + d->C1) * 2;
  const long long zb = (((long long)d->B * oh * ow - 1) * ld_dz + d->C2) * 2;
  if (xb >= 0x7fffffffLL || zb >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "wgrad: tensor exceeds 2^31 bytes");
  Y5WgradParams p{};
  p.x = x; p.dz = dz; p.dw = dw_packed; p.x_bytes = (unsigned)xb; p.dz_bytes = (unsigned)zb;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C1 = d->C1; p.ldx = d->ldx; p.OH = oh; p.OW = ow; p.C2 = d->C2; p.ldz = ld_dz;
  p.KH = d->KH; p.KW = d->KW; p.SH = d->SH; p.SW = d->SW; p.PH = d->PH; p.PW = d->PW;
  p.K = d->KH * d->KW * d->C1; p.Kpad = d->Kpad; p.Npad = d->Npad;
  if (p.Kpad < p.K || p.Npad < p.C2) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: bad packed filter dims");
  p.M = d->B * oh * ow;
  const int ncu = y5_num_cu();
  const bool det = ws != nullptr;
  // d->cfg selects the kernel family: -1 / 0 = automatic (by pixels per filter element), 1 = the general im2col-gather kernel, 3 / 3xy = the patch-staged 3x3
  // kernel (wgrad3.h).  Which one is faster depends on pixels per filter element (measured, profiles/r03/r03_wgrad3_ab.log): TrainEngine times both.
  const bool can3 = d->KH == 3 && d->KW == 3 && d->PH == 1 && d->PW == 1 && d->SH == d->SW && (d->SH == 1 || d->SH == 2) && d->C2 * 9LL * d->C1 > 0;
  // cfg 6: the stem kernel (wgrad3.h): k(6,3) s(2,1) p(2,1) on the 8-channel paired-pixel view of the 3-channel image, contiguous pixels (ldx == 8)
  const bool can6 = d->KH == 6 && d->KW == 3 && d->SH == 2 && d->SW == 1 && d->PH == 2 && d->PW == 1 && d->C1 == 8 && d->ldx == 8 && p.Kpad >= 144;
  if (d->cfg == 6 && !can6) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: cfg 6 (stem kernel) needs the paired-pixel stem geometry");
  if (can6 && (d->cfg == 6 || d->cfg <= 0)) {
    static const int variant = [] { const char* e = getenv("Y5_WG_STEM"); return e ? atoi(e) : 26; }();   // experiment knob: <segments per chunk><ring depth>
    const int NSEG = variant / 10;
    p.tiles_n = (d->C2 + 31) / 32;
    p.tiles_k = 1;
    const long long segs = (long long)d->B * oh * ((ow + 15) / 16);
    const long long chunks = (segs + NSEG - 1) / NSEG;
    long long splits = d->max_blocks > 0 ? d->max_blocks : (3 * ncu + p.tiles_n - 1) / p.tiles_n;
    const long long max_splits = (chunks + 3) / 4;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const long long cps = (chunks + splits - 1) / splits;
    p.pix_per_split = (int)(NSEG * cps);
    p.splits = (int)((chunks + cps - 1) / cps);
    const long long grid = (long long)p.tiles_n * p.splits;
    const size_t ws_need = (size_t)p.splits * p.Npad * p.Kpad * sizeof(float);
    if (need) { *need = ws_need; return Y5_OK; }
    if (det && (ws_bytes < ws_need || ((uintptr_t)ws & 15))) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: workspace too small (y5_conv2d_wgrad_ws_bytes) or misaligned");
    p.ws = ws;
#define Y5_STEM(NS, SS)                                                                                                            \
  case NS * 10 + SS: {                                                                                                             \
    constexpr size_t bytes = (size_t)SS * (NS * 16 * 64 + ((NS * 6 * 20 * 16 + 1023) / 1024) * 1024 + 64);                         \
    static bool attr = false;                                                                                                      \
    if (!attr) {                                                                                                                   \
      hipFuncSetAttribute(reinterpret_cast<const void*>(y5_conv_wgrad_stem_kernel<NS, SS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      hipFuncSetAttribute(reinterpret_cast<const void*>(y5_conv_wgrad_stem_kernel<NS, SS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
      attr = true;                                                                                                                 \
    }                                                                                                                              \
    if (det) hipLaunchKernelGGL((y5_conv_wgrad_stem_kernel<NS, SS, true>), dim3((unsigned)grid), dim3(384), bytes, st, p);         \
    else hipLaunchKernelGGL((y5_conv_wgrad_stem_kernel<NS, SS, false>), dim3((unsigned)grid), dim3(384), bytes, st, p);            \
  } break;
    switch (variant) {
      Y5_STEM(2, 6) Y5_STEM(4, 4) Y5_STEM(4, 6) Y5_STEM(8, 3) Y5_STEM(8, 4)
      default: return y5_fail(Y5_ERR_BAD_ARG, "wgrad: unknown Y5_WG_STEM variant");
    }
#undef Y5_STEM
    if (det) {
      const long long total = (long long)d->C2 * p.Kpad;
      hipLaunchKernelGGL(y5_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, dw_packed, d->C2, p.K, p.Npad, p.Kpad, p.splits);
    }
    return y5_check_launch("y5_conv2d_wgrad(stem)");
  }
  const bool want3 = d->cfg == 3 || d->cfg >= 300;
  if (want3 && !can3) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: cfg 3 (patch-staged kernel) needs k3 p1 and stride 1 or 2");
  const bool auto3 = d->cfg <= 0 && (long long)p.M >= 10LL * p.K * d->C2;   // many pixels per filter element: P1-P3 of the yolov5 graphs
  if (can3 && (want3 || auto3)) {
    int nt = d->C2 > 64 ? 4 : d->C2 > 32 ? 2 : 1, ct = d->C1 > 32 ? 2 : 1;
    if (d->cfg >= 300) {  // 300 + 10 NTcap + CTcap: the channel tile is capped (the tuner's alternative shapes, e.g. 341 = up to 128 x 32 per workgroup)
      const int cn = (d->cfg - 300) / 10, cc = (d->cfg - 300) % 10;
      if ((cn != 1 && cn != 2 && cn != 4) || (cc != 1 && cc != 2)) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: cfg 3xy needs x in {1,2,4}, y in {1,2}");
      nt = nt < cn ? nt : cn;
      ct = ct < cc ? ct : cc;
    }
    p.tiles_n = (d->C2 + 32 * nt - 1) / (32 * nt);
    p.tiles_k = (d->C1 + 32 * ct - 1) / (32 * ct);
    const int tiles = p.tiles_n * p.tiles_k;
    const long long segs = (long long)d->B * oh * ((ow + 15) / 16);
    const long long chunks = (segs + 1) / 2;
    long long splits = d->max_blocks > 0 ? d->max_blocks : (2 * ncu + tiles - 1) / tiles;
    const long long max_splits = (chunks + 3) / 4;  // at least 4 chunks per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const long long cps = (chunks + splits - 1) / splits;
    p.pix_per_split = (int)(2 * cps);                // SEGMENTS per split
    p.splits = (int)((chunks + cps - 1) / cps);
    const long long grid = (long long)tiles * p.splits;
    if (grid > 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "wgrad: grid too large");
    const size_t ws_need = (size_t)p.splits * p.Npad * p.Kpad * sizeof(float);
    if (need) { *need = ws_need; return Y5_OK; }
    if (det && (ws_bytes < ws_need || ((uintptr_t)ws & 15))) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: workspace too small (y5_conv2d_wgrad_ws_bytes) or misaligned");
    p.ws = ws;
    const int key = nt * 100 + ct * 10 + d->SH;
#define Y5_WG3(NT_, CT_, ST_, S_)                                                                                      \
  case NT_ * 100 + CT_ * 10 + ST_:                                                                                     \
    if (det) launch_wgrad3<NT_, CT_, ST_, S_, true>(p, (unsigned)grid, st);                                            \
    else launch_wgrad3<NT_, CT_, ST_, S_, false>(p, (unsigned)grid, st);                                               \
    break;
    switch (key) {
      Y5_WG3(1, 1, 1, 4) Y5_WG3(1, 1, 2, 4) Y5_WG3(1, 2, 1, 4) Y5_WG3(1, 2, 2, 4)
      Y5_WG3(2, 1, 1, 4) Y5_WG3(2, 1, 2, 4) Y5_WG3(2, 2, 1, 4) Y5_WG3(2, 2, 2, 3)
      Y5_WG3(4, 1, 1, 4) Y5_WG3(4, 1, 2, 4) Y5_WG3(4, 2, 1, 4) Y5_WG3(4, 2, 2, 3)
      default: return y5_fail(Y5_ERR_RUNTIME, "wgrad: no patch-staged instantiation");
    }
#undef Y5_WG3
    if (det) {
      const long long total = (long long)d->C2 * p.Kpad;
      hipLaunchKernelGGL(y5_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, dw_packed, d->C2, p.K, p.Npad, p.Kpad, p.splits);
    }
    return y5_check_launch("y5_conv2d_wgrad(k3)");
  }
  int tnb = d->C2 >= 128 ? 2 : 1, tkb = p.K >= 128 ? 2 : 1;
  // cfg 1xy: the general kernel with its filter tile capped at 64 x output channels by 64 y k columns (x, y in {1, 2}).  Splits x filter elements
  // is the atomic traffic of a launch; at P4 / P5 (few pixels, large filters) four times as many tiles need a quarter of the splits to fill the chip.
  if ((d->cfg > 1 && d->cfg < 100 && d->cfg != 3 && d->cfg != 6) || (d->cfg >= 200 && d->cfg < 300)) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: unknown cfg");
  if (d->cfg >= 100 && d->cfg < 200) {
    const int cn = (d->cfg - 100) / 10, ck = (d->cfg - 100) % 10;
    if ((cn != 1 && cn != 2) || (ck != 1 && ck != 2)) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: cfg 1xy needs x, y in {1, 2}");
    tnb = tnb < cn ? tnb : cn;
    tkb = tkb < ck ? tkb : ck;
  }
  p.tiles_n = (d->C2 + 64 * tnb - 1) / (64 * tnb);
  p.tiles_k = (p.K + 64 * tkb - 1) / (64 * tkb);
  const int tiles = p.tiles_n * p.tiles_k;
  int splits = d->max_blocks > 0 ? d->max_blocks : (4 * ncu + tiles - 1) / tiles;
  const int max_splits = (p.M + 255) / 256;  // at least 8 chunks per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.pix_per_split = (((p.M + splits - 1) / splits) + 31) / 32 * 32;
  p.splits = (p.M + p.pix_per_split - 1) / p.pix_per_split;
  const long long grid = (long long)tiles * p.splits;
  if (grid > 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "wgrad: grid too large");
  const size_t ws_need = (size_t)p.splits * p.Npad * p.Kpad * sizeof(float);
  if (need) { *need = ws_need; return Y5_OK; }
  if (det && (ws_bytes < ws_need || ((uintptr_t)ws & 15))) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: workspace too small (y5_conv2d_wgrad_ws_bytes) or misaligned");
  p.ws = ws;
  // ring depth: as many 32-pixel chunks in flight as ~48 KiB of LDS per workgroup allows (3 workgroups per CU)
  const bool lin = d->KH == 1 && d->KW == 1 && d->SH == 1 && d->SW == 1 && d->PH == 0 && d->PW == 0;
#define Y5_WG_LAUNCH(TN, TK, SS, BYTES)                                                                                          \
  do {                                                                                                                           \
    if (det) {                                                                                                                   \
      if (lin) hipLaunchKernelGGL((y5_conv_wgrad_kernel<TN, TK, SS, true, true>), dim3((unsigned)grid), dim3(256), BYTES, st, p);  \
      else hipLaunchKernelGGL((y5_conv_wgrad_kernel<TN, TK, SS, false, true>), dim3((unsigned)grid), dim3(256), BYTES, st, p);     \
    } else if (lin) hipLaunchKernelGGL((y5_conv_wgrad_kernel<TN, TK, SS, true>), dim3((unsigned)grid), dim3(256), BYTES, st, p);   \
    else hipLaunchKernelGGL((y5_conv_wgrad_kernel<TN, TK, SS, false>), dim3((unsigned)grid), dim3(256), BYTES, st, p);            \
  } while (0)
  if (tnb == 2 && tkb == 2) Y5_WG_LAUNCH(2, 2, 3, 3 * 16384);
  else if (tnb == 2) Y5_WG_LAUNCH(2, 1, 4, 4 * 12288);
  else if (tkb == 2) Y5_WG_LAUNCH(1, 2, 4, 4 * 12288);
  else Y5_WG_LAUNCH(1, 1, 6, 6 * 8192);
#undef Y5_WG_LAUNCH
  if (det) {
    const long long total = (long long)d->C2 * p.Kpad;
    hipLaunchKernelGGL(y5_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, dw_packed, d->C2, p.K, p.Npad, p.Kpad, p.splits);
  }
  return y5_check_launch("y5_conv2d_wgrad");
}

extern "C" int y5_conv2d_wgrad(const y5_conv_desc* d, const void* x, const void* dz, int ld_dz, float* dw_packed, void* stream_) {
  return wgrad_impl(d, x, dz, ld_dz, dw_packed, nullptr, 0, nullptr, stream_);
}
extern "C" int y5_conv2d_wgrad_det(const y5_conv_desc* d, const void* x, const void* dz, int ld_dz, float* dw_packed, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
  if (!workspace) return y5_fail(Y5_ERR_BAD_ARG, "wgrad_det: null workspace");
  return wgrad_impl(d, x, dz, ld_dz, dw_packed, static_cast<float*>(workspace), workspace_bytes, nullptr, stream_);
}
extern "C" long long y5_conv2d_wgrad_ws_bytes(const y5_conv_desc* d, int ld_dz) {
  size_t need = 0;
  if (wgrad_impl(d, nullptr, nullptr, ld_dz, nullptr, nullptr, 0, &need, nullptr) != Y5_OK) return -1;
  return (long long)need;
}
