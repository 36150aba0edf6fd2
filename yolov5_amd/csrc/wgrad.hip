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

__global__ __launch_bounds__(256)
void y5_conv_wgrad_kernel(const Y5WgradParams p) {
  typedef half_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][dz tile 32 x 128 B | x tile 32 x 128 B]
  constexpr int TILE = 32 * 128, BUF = 2 * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  const int tk = bid % p.tiles_k;
  const int tn = (bid / p.tiles_k) % p.tiles_n;
  const int sp = bid / (p.tiles_k * p.tiles_n);
  const int n0 = tn * 64, k0 = tk * 64;
  const int m_begin = sp * p.pix_per_split;
  const int m_end = m_begin + p.pix_per_split < p.M ? m_begin + p.pix_per_split : p.M;
  if (m_begin >= m_end) return;
  const int nchunks = (m_end - m_begin + 31) / 32;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t zrs = y5_make_rsrc(p.dz, p.dz_bytes);

  // staging roles: 8 LDS-DMA instructions per chunk (4 dz + 4 x), 2 per wave; instruction I covers rows 8*(I&3) .. +7
  const int lrow = lane >> 3, lslot = lane & 7;
  // x piece of this lane: k = k0 + 8*lslot -> tap, channel
  const int kx = k0 + 8 * lslot;
  const bool kx_ok = kx < p.K;
  const int tap = kx_ok ? kx / p.C1 : 0;
  const int cx = kx - tap * p.C1;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const bool nz_ok = n0 + 8 * lslot < p.Npad && n0 + 8 * lslot < p.C2 + 7;  // dz columns beyond C2 are never used
  const int ohw = p.OH * p.OW;

  auto stage = [&](int ch, int buf) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int I = wave * 2 + q;            // 0..7
      const int r = (I & 3) * 8 + lrow;      // pixel row inside the chunk
      const int m = m_begin + ch * 32 + r;
      unsigned voff = Y5_OOB;
      if (I < 4) {                           // dz rows
        if (m < m_end && nz_ok) voff = (unsigned)((m * p.ldz + n0 + 8 * lslot) * 2);
        y5_bglds16(zrs, voff, base + (I & 3) * 1024);
      } else {                               // gathered x rows
        if (m < m_end && kx_ok) {
          const int b = m / ohw;
          const int rr = m - b * ohw;
          const int oh = rr / p.OW, ow = rr - oh * p.OW;
          const int ih = oh * p.SH - p.PH + kh, iw = ow * p.SW - p.PW + kw;
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
            voff = (unsigned)((((b * p.H + ih) * p.W + iw) * p.ldx + cx) * 2);
        }
        y5_bglds16(xrs, voff, base + TILE + (I & 3) * 1024);
      }
    }
  };

  // MFMA roles: wave -> 32 x 32 sub-tile (wn, wk) of the 64 x 64 block tile
  const int wn = wave >> 1, wk = wave & 1;
  const int fi = lane & 31, g = lane >> 5;
  float16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cur = ch & 1;
    if (ch + 1 < nchunks) stage(ch + 1, cur ^ 1);
    const T* zt = reinterpret_cast<const T*>(smem + cur * BUF) + wn * 32 + fi;          // [p][64]: column n
    const T* xt = reinterpret_cast<const T*>(smem + cur * BUF + TILE) + wk * 32 + fi;   // [p][64]: column k
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8_t af, bf;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int pr = ks * 16 + g * 8 + e;
        af[e] = zt[pr * 64];
        bf[e] = xt[pr * 64];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  // D[i][j]: col j = lane & 31 (k), row i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (n)
  const int kcol = k0 + wk * 32 + fi;
  if (kcol < p.K) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (n < p.C2) atomicAdd(p.dw + (size_t)n * p.Kpad + kcol, acc[r]);
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
  p.tiles_n = (d->C2 + 63) / 64;
  p.tiles_k = (p.K + 63) / 64;
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
  hipLaunchKernelGGL(y5_conv_wgrad_kernel, dim3((unsigned)grid), dim3(256), 2 * 2 * 32 * 128, st, p);
  return y5_check_launch("y5_conv2d_wgrad");
}
