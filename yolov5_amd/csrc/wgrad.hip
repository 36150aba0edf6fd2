// Weight gradient of the convolution (autograd of models/common.py:74-92 `Conv`, train.py:410 `scaler.scale(loss).backward()`):
//     dW[n][kh][kw][c] = sum over (b, oh, ow) of dz[b, oh, ow, n] * x[b, oh*s + kh - p, ow*s + kw - p, c]
// written into the packed [Npad][Kpad] filter layout of y5_conv2d_fwd (k = (kh, kw, c)), fp32, accumulated with atomics
// (the caller zero-fills dW).  GEMM view: D[n][k] += sum_p dzT[n][p] * X[p][k] with the contraction over pixels, so both
// MFMA operands need 8 consecutive PIXELS of one channel per lane -- a transposed read of the pixel-major NHWC tensors:
// 32-pixel chunks of dz (64 output channels) and of the gathered x columns (64 k) are staged row-major in LDS by
// buffer-addressed LDS-DMA (image borders / tails = zero fill) and the fragments are gathered with 16-bit LDS reads.
// One workgroup = one 64 x 64 tile of dW and one slice of the pixel range (split-K over the grid).
// v1 of this kernel is LDS-issue bound (16 ds_read_u16 per MFMA); DESIGN.md lists the ds_read_b64_tr_b16 upgrade.
#include <hip/hip_runtime.h>

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
};

template <int TNB, int TKB>
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
  int i_isx[NI], i_row[NI], i_col8[NI], i_dst[NI];   // per instruction of this wave: kind, pixel row, 8-channel group, LDS offset
  int x_kh[NI], x_kw[NI], x_c[NI];
  bool i_ok[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int I = wave * NI + q;
    const bool isx = I >= ZI;
    const int J = isx ? I - ZI : I;
    const int rowb = isx ? XROW : ZROW;
    const int lpr = rowb / 16;                       // lanes per pixel row
    const int r = J * (1024 / rowb) + lane / lpr;    // pixel row inside the chunk
    const int s8 = lane % lpr;                       // 16-byte group inside the row
    i_isx[q] = isx; i_row[q] = r; i_col8[q] = s8; i_dst[q] = (isx ? ZT : 0) + J * 1024;
    if (isx) {
      const int kx = k0 + 8 * s8;
      i_ok[q] = kx < p.K;
      const int tap = i_ok[q] ? kx / p.C1 : 0;
      x_c[q] = kx - tap * p.C1;
      x_kh[q] = tap / p.KW; x_kw[q] = tap - x_kh[q] * p.KW;
    } else {
      i_ok[q] = n0 + 8 * s8 < p.C2;
      x_c[q] = x_kh[q] = x_kw[q] = 0;
    }
  }

  auto stage = [&](int ch, int buf) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int m = m_begin + ch * 32 + i_row[q];
      unsigned voff = Y5_OOB;
      if (m < m_end && i_ok[q]) {
        if (!i_isx[q]) {
          voff = (unsigned)((m * p.ldz + n0 + 8 * i_col8[q]) * 2);
        } else {
          const int b = m / ohw;
          const int rr = m - b * ohw;
          const int oh = rr / p.OW, ow = rr - oh * p.OW;
          const int ih = oh * p.SH - p.PH + x_kh[q], iw = ow * p.SW - p.PW + x_kw[q];
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
            voff = (unsigned)((((b * p.H + ih) * p.W + iw) * p.ldx + x_c[q]) * 2);
        }
      }
      y5_bglds16(i_isx[q] ? xrs : zrs, voff, base + i_dst[q]);
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

  stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cur = ch & 1;
    if (ch + 1 < nchunks) stage(ch + 1, cur ^ 1);
    const T* zt = reinterpret_cast<const T*>(smem + cur * BUF) + wn * 32 * TNB + fi;        // [p][64*TNB]: column n
    const T* xt = reinterpret_cast<const T*>(smem + cur * BUF + ZT) + wk * 32 * TKB + fi;   // [p][64*TKB]: column k
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8_t af[TNB], bf[TKB];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int pr = ks * 16 + g * 8 + e;
#pragma unroll
        for (int a = 0; a < TNB; ++a) af[a][e] = zt[pr * (64 * TNB) + a * 32];
#pragma unroll
        for (int b2 = 0; b2 < TKB; ++b2) bf[b2][e] = xt[pr * (64 * TKB) + b2 * 32];
      }
#pragma unroll
      for (int a = 0; a < TNB; ++a)
#pragma unroll
        for (int b2 = 0; b2 < TKB; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b2], acc[a][b2], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
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
        if (n < p.C2) atomicAdd(p.dw + (size_t)n * p.Kpad + kcol, acc[a][b2][r]);
      }
  }
}

extern "C" int y5_conv2d_wgrad(const y5_conv_desc* d, const void* x, const void* dz, int ld_dz, float* dw_packed, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!d || !x || !dz || !dw_packed) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: null pointer");
  if (d->dtype != Y5_F16) return y5_fail(Y5_ERR_UNSUPPORTED, "wgrad: fp16 activations / gradients only");
  if (d->C1 % 8 || d->ldx % 8 || ld_dz % 8 || d->C2 % 8) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: C1, C2, ldx, ld_dz must be multiples of 8");
  const int oh = (d->H + 2 * d->PH - d->KH) / d->SH + 1, ow = (d->W + 2 * d->PW - d->KW) / d->SW + 1;
  if (oh != d->OH || ow != d->OW) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: OH/OW inconsistent with H/W/k/s/p");
  if (((uintptr_t)x | (uintptr_t)dz | (uintptr_t)dw_packed) & 15) return y5_fail(Y5_ERR_BAD_ARG, "wgrad: pointers must be 16-byte aligned");
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
  const int tnb = d->C2 >= 128 ? 2 : 1, tkb = p.K >= 128 ? 2 : 1;
  p.tiles_n = (d->C2 + 64 * tnb - 1) / (64 * tnb);
  p.tiles_k = (p.K + 64 * tkb - 1) / (64 * tkb);
  int ncu = 256;
  {
    int dev = 0, n = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
  }
  const int tiles = p.tiles_n * p.tiles_k;
  int splits = d->max_blocks > 0 ? d->max_blocks : (4 * ncu + tiles - 1) / tiles;
  const int max_splits = (p.M + 255) / 256;  // at least 8 chunks per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.pix_per_split = (((p.M + splits - 1) / splits) + 31) / 32 * 32;
  p.splits = (p.M + p.pix_per_split - 1) / p.pix_per_split;
  const long long grid = (long long)tiles * p.splits;
  if (grid > 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "wgrad: grid too large");
  const size_t lds = (size_t)2 * 32 * 128 * (tnb + tkb);
  if (tnb == 2 && tkb == 2) hipLaunchKernelGGL((y5_conv_wgrad_kernel<2, 2>), dim3((unsigned)grid), dim3(256), lds, st, p);
  else if (tnb == 2) hipLaunchKernelGGL((y5_conv_wgrad_kernel<2, 1>), dim3((unsigned)grid), dim3(256), lds, st, p);
  else if (tkb == 2) hipLaunchKernelGGL((y5_conv_wgrad_kernel<1, 2>), dim3((unsigned)grid), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((y5_conv_wgrad_kernel<1, 1>), dim3((unsigned)grid), dim3(256), lds, st, p);
  return y5_check_launch("y5_conv2d_wgrad");
}
