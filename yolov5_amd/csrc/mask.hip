// process_mask / crop_mask (utils/segment/general.py:10-51) for gfx950 as ONE kernel per image:
//   masks = sigmoid(coef (n,c) @ proto (c, mh*mw)) -> zero outside box * (mw/iw, mh/ih) -> bilinear upsample to
//   (ih, iw) (align_corners=False) -> > 0.5
// One workgroup per (instance, 64x64 output tile).  The (TS+2)^2 low-resolution mask values the tile's bilinear taps
// can touch are computed once into LDS (c-term dot product against the prototype planes, coalesced along x), then
// every lane produces 4 consecutive output pixels and stores them as one vector.  The kernel is HBM-write bound:
// n*ih*iw output bytes (uint8) or 4x that (float32, the reference's return type).
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "y5_common.h"
#include "y5_host.h"

struct MaskParams {
  const void* protos;   // (c, mh, mw)
  const float* coef;    // row i at coef + i*ld_m, c values
  const float* boxes;   // row i at boxes + i*ld_b: x1,y1,x2,y2 in input-image pixels
  void* out;            // (n, oh, ow)
  int c, mh, mw, ld_m, ld_b, n, ih, iw, oh, ow, upsample;
  float sx, sy;         // mw/iw, mh/ih as fp32 (general.py:43-46)
  float rw, rh;         // source-per-destination scale of the bilinear resize = mw/ow, mh/oh
};

static constexpr int TILE = 64;      // output tile edge
static constexpr int MAXSRC = 68;    // low-res window edge upper bound for LDS (tile 64 at scale 1 + 2)

template <typename TP, typename TO>
__global__ __launch_bounds__(256)
void y5_process_mask_kernel(const MaskParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_m = reinterpret_cast<float*>(smem);                 // [wh][ww] cropped sigmoid masks
  float* s_coef = s_m + MAXSRC * MAXSRC;                       // [c]
  const int inst = blockIdx.y;
  const int tiles_x = (p.ow + TILE - 1) / TILE;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int X0 = tx * TILE, Y0 = ty * TILE;
  const int tid = threadIdx.x;
  // low-resolution window covering the bilinear taps of output rows [Y0, Y0+TILE) / cols [X0, X0+TILE)
  int wx0, wy0, ww, wh;
  if (p.upsample) {
    const float fx0 = fmaxf(p.rw * ((float)X0 + 0.5f) - 0.5f, 0.f), fy0 = fmaxf(p.rh * ((float)Y0 + 0.5f) - 0.5f, 0.f);
    const int Xl = (X0 + TILE < p.ow ? X0 + TILE : p.ow) - 1, Yl = (Y0 + TILE < p.oh ? Y0 + TILE : p.oh) - 1;
    const float fx1 = fmaxf(p.rw * ((float)Xl + 0.5f) - 0.5f, 0.f), fy1 = fmaxf(p.rh * ((float)Yl + 0.5f) - 0.5f, 0.f);
    wx0 = (int)fx0; wy0 = (int)fy0;
    int wx1 = (int)fx1 + 1, wy1 = (int)fy1 + 1;
    wx1 = wx1 > p.mw - 1 ? p.mw - 1 : wx1;
    wy1 = wy1 > p.mh - 1 ? p.mh - 1 : wy1;
    ww = wx1 - wx0 + 1; wh = wy1 - wy0 + 1;
  } else {
    wx0 = X0; wy0 = Y0;
    ww = p.mw - X0 < TILE ? p.mw - X0 : TILE;
    wh = p.mh - Y0 < TILE ? p.mh - Y0 : TILE;
  }
  const float* bx = p.boxes + (long long)inst * p.ld_b;
  const float x1 = bx[0] * p.sx, y1 = bx[1] * p.sy, x2 = bx[2] * p.sx, y2 = bx[3] * p.sy;  // general.py:42-46
  for (int i = tid; i < p.c; i += 256) s_coef[i] = p.coef[(long long)inst * p.ld_m + i];
  __syncthreads();
  const TP* P = static_cast<const TP*>(p.protos);
  const long long plane = (long long)p.mh * p.mw;
  for (int i = tid; i < ww * wh; i += 256) {
    const int ly = i / ww, lx = i - ly * ww;
    const int gy = wy0 + ly, gx = wx0 + lx;
    const float r = (float)gx, cc = (float)gy;
    float v = 0.f;
    if (r >= x1 && r < x2 && cc >= y1 && cc < y2) {  // crop_mask, general.py:22
      float s = 0.f;
      const TP* q = P + (long long)gy * p.mw + gx;
      int k = 0;
      for (; k + 8 <= p.c; k += 8) {   // eight prototype planes in flight (round 6; same order of additions)
        TP t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = q[(k + e) * plane];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += s_coef[k + e] * (float)t[e];
      }
      for (; k < p.c; ++k) s += s_coef[k] * (float)q[k * plane];
      v = 1.0f / (1.0f + expf(-s));
    }
    s_m[i] = v;
  }
  __syncthreads();
  TO* out = static_cast<TO*>(p.out) + (long long)inst * p.oh * p.ow;
  typedef TO TO4 __attribute__((ext_vector_type(4)));
  const bool vec_ok = (p.ow & 3) == 0;
  for (int i = tid; i < TILE * (TILE / 4); i += 256) {
    const int oy = i / (TILE / 4), ox4 = (i - oy * (TILE / 4)) * 4;
    const int Y = Y0 + oy, X = X0 + ox4;
    if (Y >= p.oh || X >= p.ow) continue;
    TO r4[4];
    float hl1 = 0.f;
    int h1 = Y - wy0, h1p = 0;
    if (p.upsample) {  // upsample_bilinear2d, align_corners=False (F.interpolate, general.py:50)
      const float h1r = fmaxf(p.rh * ((float)Y + 0.5f) - 0.5f, 0.f);
      const int hh = (int)h1r;
      h1p = hh < p.mh - 1 ? 1 : 0;
      hl1 = h1r - (float)hh;
      h1 = hh - wy0;
    }
    const float hl0 = 1.0f - hl1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int Xe = X + e;
      float val = 0.f;
      if (Xe < p.ow) {
        if (p.upsample) {
          const float w1r = fmaxf(p.rw * ((float)Xe + 0.5f) - 0.5f, 0.f);
          const int w1i = (int)w1r;
          const int w1p = w1i < p.mw - 1 ? 1 : 0;
          const float wl1 = w1r - (float)w1i, wl0 = 1.0f - wl1;
          const int w1 = w1i - wx0;
          const float v00 = s_m[h1 * ww + w1], v01 = s_m[h1 * ww + w1 + w1p];
          const float v10 = s_m[(h1 + h1p) * ww + w1], v11 = s_m[(h1 + h1p) * ww + w1 + w1p];
          val = hl0 * (wl0 * v00 + wl1 * v01) + hl1 * (wl0 * v10 + wl1 * v11);
        } else {
          val = s_m[h1 * ww + (Xe - wx0)];
        }
      }
      r4[e] = (TO)(val > 0.5f ? 1 : 0);  // general.py:51 gt_(0.5)
    }
    TO* d = out + (long long)Y * p.ow + X;
    if (vec_ok && X + 3 < p.ow) {
      TO4 v4; v4[0] = r4[0]; v4[1] = r4[1]; v4[2] = r4[2]; v4[3] = r4[3];
      *reinterpret_cast<TO4*>(d) = v4;
    } else {
      for (int e = 0; e < 4 && X + e < p.ow; ++e) d[e] = r4[e];
    }
  }
}


// ---- the whole batch in ONE launch (round 6) -----------------------------------------------------------------------------------------------------------
// segment/predict.py:161-172 calls process_mask once per image; at bs = 32 with 300 instances each that is 32 launches of 30 000 workgroups apiece.  Here ONE
// grid covers the (image, instance, tile) items of the WHOLE batch; a tile is 64 output rows x 256 BYTES of a row (64 float32 / 256 uint8 pixels: every row
// segment is 16 lanes x 16 bytes).  Most tiles lie wholly outside their instance's box (crop_mask zeroes everything outside): those are pure zero stores -- no
// LDS, no barrier, no prototype read.  Arithmetic of a non-zero tile is y5_process_mask_kernel's, expression for expression.
// (A first version walked full-width 32-row strips -- one contiguous 80 KB run per workgroup: 0.27 TB/s in float32, because concurrent workgroups then write
// at a 20 x 4 KB stride and camp on a quarter of the memory channels; the same kernel wrote uint8 (stride 5 x 4 KB) at 1.5 TB/s.  profiles/r06/r06_mask_batch.log)
struct MaskImg { const float* coef; const float* boxes; int ld_m, ld_b, n, out_off; };
static constexpr int MB_MAX_IMG = 64;
struct MaskBatchParams {
  const void* protos;   // (B, c, mh, mw)
  void* out;            // (sum n, oh, ow)
  MaskImg img[MB_MAX_IMG];
  int B, c, mh, mw, ih, iw, oh, ow, upsample, total, tiles_x, tiles_y;
  float sx, sy, rw, rh;
};
static constexpr int MB_TH = 64;   // output rows per tile

template <typename TP, typename TO>
__global__ __launch_bounds__(256)
void y5_process_mask_batch_kernel(const MaskBatchParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = 16 / (int)sizeof(TO);
  constexpr int TW = 16 * VEC;                                  // output columns per tile: 256 bytes of a row
  constexpr int WMAX_X = TW + 4, WMAX_Y = MB_TH + 4;            // low-resolution window bounds (scale 1 = no upsampling)
  float* s_m = reinterpret_cast<float*>(smem);                  // [wh][ww] cropped sigmoid masks
  float* s_coef = s_m + WMAX_X * WMAX_Y;                        // [c]
  const int tid = threadIdx.x;
  const int per_inst = p.tiles_x * p.tiles_y;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const int vy = tid >> 4, vx = (tid & 15) * VEC;               // this lane's vector inside a 16-row slab of the tile
  {
    // grid: x = (instance slot, tile) up to the LARGEST image of the launch, y = image -- no descriptor search, and the hardware's workgroup scheduler overlaps
    // the latency chains of many items (a persistent grid of 8 workgroups per CU ran the items of a workgroup one after the other: box load -> prototype loads
    // -> stores, ~30 us per non-zero tile, 0.9 TB/s in uint8)
    const int b = blockIdx.y;
    const MaskImg im = p.img[b];
    const int inst = blockIdx.x / per_inst, ti = blockIdx.x - inst * per_inst;
    if (inst >= im.n) return;
    const int g = im.out_off + inst;
    const int ty = ti / p.tiles_x, tx = ti - ty * p.tiles_x;
    const int X0 = tx * TW, Y0 = ty * MB_TH;
    // low-resolution window covering the bilinear taps of output rows [Y0, Y0 + TH) / columns [X0, X0 + TW): y5_process_mask_kernel's
    int wx0, wy0, ww, wh;
    if (p.upsample) {
      const float fx0 = fmaxf(p.rw * ((float)X0 + 0.5f) - 0.5f, 0.f), fy0 = fmaxf(p.rh * ((float)Y0 + 0.5f) - 0.5f, 0.f);
      const int Xl = (X0 + TW < p.ow ? X0 + TW : p.ow) - 1, Yl = (Y0 + MB_TH < p.oh ? Y0 + MB_TH : p.oh) - 1;
      const float fx1 = fmaxf(p.rw * ((float)Xl + 0.5f) - 0.5f, 0.f), fy1 = fmaxf(p.rh * ((float)Yl + 0.5f) - 0.5f, 0.f);
      wx0 = (int)fx0; wy0 = (int)fy0;
      int wx1 = (int)fx1 + 1, wy1 = (int)fy1 + 1;
      wx1 = wx1 > p.mw - 1 ? p.mw - 1 : wx1;
      wy1 = wy1 > p.mh - 1 ? p.mh - 1 : wy1;
      ww = wx1 - wx0 + 1; wh = wy1 - wy0 + 1;
    } else {
      wx0 = X0; wy0 = Y0;
      ww = p.mw - X0 < TW ? p.mw - X0 : TW;
      wh = p.mh - Y0 < MB_TH ? p.mh - Y0 : MB_TH;
    }
    const float* bx = im.boxes + (long long)inst * im.ld_b;
    const float x1 = bx[0] * p.sx, y1 = bx[1] * p.sy, x2 = bx[2] * p.sx, y2 = bx[3] * p.sy;  // general.py:42-46
    // does any low-resolution pixel of the window lie inside the crop (x1 <= gx < x2, y1 <= gy < y2; general.py:22)?  Evaluated with the SAME float
    // comparisons on the window's corner pixels as the per-pixel test below (monotone in gx / gy), so an all-zero window is exactly an all-zero tile.
    const float gxa = (float)wx0, gxb = (float)(wx0 + ww - 1), gya = (float)wy0, gyb = (float)(wy0 + wh - 1);
    const bool hit = gxb >= x1 && gxa < x2 && gyb >= y1 && gya < y2 && x1 < x2 && y1 < y2;
    TO* out = static_cast<TO*>(p.out) + (long long)g * p.oh * p.ow;
    if (!hit) {
      const u4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < MB_TH / 16; ++k) {
        const int Y = Y0 + k * 16 + vy, X = X0 + vx;
        if (Y < p.oh && X < p.ow) *reinterpret_cast<u4*>(out + (long long)Y * p.ow + X) = z;
      }
      return;
    }
    for (int i = tid; i < p.c; i += 256) s_coef[i] = im.coef[(long long)inst * im.ld_m + i];
    __syncthreads();
    const TP* P = static_cast<const TP*>(p.protos) + (long long)b * p.c * p.mh * p.mw;
    const long long plane = (long long)p.mh * p.mw;
    for (int i = tid; i < ww * wh; i += 256) {
      const int ly = i / ww, lx = i - ly * ww;
      const int gy = wy0 + ly, gx = wx0 + lx;
      const float r = (float)gx, cc = (float)gy;
      float v = 0.f;
      if (r >= x1 && r < x2 && cc >= y1 && cc < y2) {  // crop_mask, general.py:22
        float s = 0.f;
        const TP* q = P + (long long)gy * p.mw + gx;
        int k = 0;
        for (; k + 8 <= p.c; k += 8) {   // eight prototype planes in flight (the plain loop waits for every load before the next: a latency chain of c round trips)
          TP t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = q[(k + e) * plane];
#pragma unroll
          for (int e = 0; e < 8; ++e) s += s_coef[k + e] * (float)t[e];   // (same order of additions as the plain loop)
        }
        for (; k < p.c; ++k) s += s_coef[k] * (float)q[k * plane];
        v = 1.0f / (1.0f + expf(-s));
      }
      s_m[i] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < MB_TH / 16; ++k) {
      const int Y = Y0 + k * 16 + vy, X = X0 + vx;
      if (Y >= p.oh || X >= p.ow) continue;
      TO r[VEC];
      float hl1 = 0.f;
      int h1 = Y - wy0, h1p = 0;
      if (p.upsample) {  // upsample_bilinear2d, align_corners=False (F.interpolate, general.py:50)
        const float h1r = fmaxf(p.rh * ((float)Y + 0.5f) - 0.5f, 0.f);
        const int hh = (int)h1r;
        h1p = hh < p.mh - 1 ? 1 : 0;
        hl1 = h1r - (float)hh;
        h1 = hh - wy0;
      }
      const float hl0 = 1.0f - hl1;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int Xe = X + e;   // (ow % VEC == 0: a vector never straddles the row end)
        float val;
        if (p.upsample) {
          const float w1r = fmaxf(p.rw * ((float)Xe + 0.5f) - 0.5f, 0.f);
          const int w1i = (int)w1r;
          const int w1p = w1i < p.mw - 1 ? 1 : 0;
          const float wl1 = w1r - (float)w1i, wl0 = 1.0f - wl1;
          const int w1 = w1i - wx0;
          const float v00 = s_m[h1 * ww + w1], v01 = s_m[h1 * ww + w1 + w1p];
          const float v10 = s_m[(h1 + h1p) * ww + w1], v11 = s_m[(h1 + h1p) * ww + w1 + w1p];
          val = hl0 * (wl0 * v00 + wl1 * v01) + hl1 * (wl0 * v10 + wl1 * v11);
        } else {
          val = s_m[h1 * ww + (Xe - wx0)];
        }
        r[e] = (TO)(val > 0.5f ? 1 : 0);  // general.py:51 gt_(0.5)
      }
      u4 v4;
      __builtin_memcpy(&v4, r, 16);
      *reinterpret_cast<u4*>(out + (long long)Y * p.ow + X) = v4;
    }
  }
}

extern "C" int y5_process_mask_batch(const void* protos, int proto_dtype, int B, int c, int mh, int mw, const y5_mask_img* imgs, int ih, int iw,
                                     int upsample, void* out, int out_dtype, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!protos || !imgs || B < 1 || c < 1 || c > 256 || mh < 1 || mw < 1 || ih < 1 || iw < 1) return y5_fail(Y5_ERR_BAD_ARG, "process_mask_batch: bad args");
  if (proto_dtype != Y5_F16 && proto_dtype != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "process_mask_batch: protos must be f16 or f32");
  if (out_dtype != Y5_F32 && out_dtype != Y5_U8) return y5_fail(Y5_ERR_BAD_ARG, "process_mask_batch: out dtype must be Y5_F32 or Y5_U8");
  const int oh = upsample ? ih : mh, ow = upsample ? iw : mw;
  const int vec = out_dtype == Y5_F32 ? 4 : 16;
  const float rw = (float)mw / (float)ow, rh = (float)mh / (float)oh;
  if (upsample && (rw > 1.0f || rh > 1.0f)) return y5_fail(Y5_ERR_UNSUPPORTED, "process_mask_batch: only upsampling (ih >= mh, iw >= mw) is supported");
  const int tw = 16 * vec;
  const size_t lds = ((size_t)(tw + 4) * (MB_TH + 4) + 256) * 4;
  if (ow % vec || ((uintptr_t)out & 15)) return y5_fail(Y5_ERR_UNSUPPORTED, "process_mask_batch: needs ow % (16 bytes) == 0 and a 16-byte aligned output (use y5_process_mask)");
  long long done = 0;   // instances of the images of earlier chunks
  const size_t esz = out_dtype == Y5_F32 ? 4 : 1;
  for (int b0 = 0; b0 < B; b0 += MB_MAX_IMG) {
    MaskBatchParams p{};
    const int nb = B - b0 < MB_MAX_IMG ? B - b0 : MB_MAX_IMG;
    int total = 0;
    for (int i = 0; i < nb; ++i) {
      const y5_mask_img& s = imgs[b0 + i];
      if (s.n < 0 || (s.n > 0 && (!s.masks_in || !s.boxes || s.ld_m < c || s.ld_b < 4))) return y5_fail(Y5_ERR_BAD_ARG, "process_mask_batch: bad image descriptor");
      p.img[i] = MaskImg{s.masks_in, s.boxes, s.ld_m, s.ld_b, s.n, total};
      total += s.n;
    }
    if (total > 0) {
      if (!out) return y5_fail(Y5_ERR_BAD_ARG, "process_mask_batch: null output");
      p.protos = static_cast<const char*>(protos) + (size_t)b0 * c * mh * mw * (proto_dtype == Y5_F16 ? 2 : 4);
      p.out = static_cast<char*>(out) + (size_t)done * oh * ow * esz;
      p.B = nb; p.c = c; p.mh = mh; p.mw = mw; p.ih = ih; p.iw = iw; p.oh = oh; p.ow = ow; p.upsample = upsample ? 1 : 0;
      p.total = total; p.tiles_x = (ow + tw - 1) / tw; p.tiles_y = (oh + MB_TH - 1) / MB_TH;
      p.sx = (float)((double)mw / (double)iw); p.sy = (float)((double)mh / (double)ih);
      p.rw = rw; p.rh = rh;
      int nmax = 0;
      for (int i = 0; i < nb; ++i) nmax = p.img[i].n > nmax ? p.img[i].n : nmax;
      const long long gx = (long long)nmax * p.tiles_x * p.tiles_y;
      if (gx > 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "process_mask_batch: grid out of range");
      const dim3 grid((unsigned)gx, (unsigned)nb), block(256);
      if (proto_dtype == Y5_F16) {
        if (out_dtype == Y5_F32) hipLaunchKernelGGL((y5_process_mask_batch_kernel<half_t, float>), grid, block, lds, st, p);
        else hipLaunchKernelGGL((y5_process_mask_batch_kernel<half_t, unsigned char>), grid, block, lds, st, p);
      } else {
        if (out_dtype == Y5_F32) hipLaunchKernelGGL((y5_process_mask_batch_kernel<float, float>), grid, block, lds, st, p);
        else hipLaunchKernelGGL((y5_process_mask_batch_kernel<float, unsigned char>), grid, block, lds, st, p);
      }
      const int rc = y5_check_launch("y5_process_mask_batch");
      if (rc) return rc;
    }
    done += total;
  }
  return Y5_OK;
}

extern "C" int y5_process_mask(const void* protos, int proto_dtype, int c, int mh, int mw, const float* masks_in, int ld_m,
                               const float* boxes, int ld_b, int n, int ih, int iw, int upsample, void* out, int out_dtype,
                               void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (n == 0) return Y5_OK;
  if (!protos || !masks_in || !boxes || !out || n < 0 || c < 1 || c > 256 || mh < 1 || mw < 1 || ih < 1 || iw < 1 || ld_m < c || ld_b < 4)
    return y5_fail(Y5_ERR_BAD_ARG, "process_mask: bad args");
  if (proto_dtype != Y5_F16 && proto_dtype != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "process_mask: protos must be f16 or f32");
  if (out_dtype != Y5_F32 && out_dtype != Y5_U8) return y5_fail(Y5_ERR_BAD_ARG, "process_mask: out dtype must be Y5_F32 or Y5_U8");
  MaskParams p{};
  p.protos = protos; p.coef = masks_in; p.boxes = boxes; p.out = out;
  p.c = c; p.mh = mh; p.mw = mw; p.ld_m = ld_m; p.ld_b = ld_b; p.n = n; p.ih = ih; p.iw = iw; p.upsample = upsample ? 1 : 0;
  p.oh = upsample ? ih : mh; p.ow = upsample ? iw : mw;
  p.sx = (float)((double)mw / (double)iw); p.sy = (float)((double)mh / (double)ih);
  p.rw = (float)mw / (float)p.ow; p.rh = (float)mh / (float)p.oh;
  if (upsample && (p.rw > 1.0f || p.rh > 1.0f)) return y5_fail(Y5_ERR_UNSUPPORTED, "process_mask: only upsampling (ih >= mh, iw >= mw) is supported");
  const int tiles = ((p.ow + TILE - 1) / TILE) * ((p.oh + TILE - 1) / TILE);
  const dim3 grid((unsigned)tiles, (unsigned)n), block(256);
  const size_t lds = (size_t)(MAXSRC * MAXSRC + 256) * 4;
  if (proto_dtype == Y5_F16) {
    if (out_dtype == Y5_F32) hipLaunchKernelGGL((y5_process_mask_kernel<half_t, float>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((y5_process_mask_kernel<half_t, unsigned char>), grid, block, lds, st, p);
  } else {
    if (out_dtype == Y5_F32) hipLaunchKernelGGL((y5_process_mask_kernel<float, float>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((y5_process_mask_kernel<float, unsigned char>), grid, block, lds, st, p);
  }
  return y5_check_launch("y5_process_mask");
}
