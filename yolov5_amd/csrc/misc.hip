// C-ABI launchers for the HBM-bound glue kernels (misc_kernels.h).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/yolov5_hip.h"
#include "misc_kernels.h"
#include "y5_host.h"

namespace {
inline unsigned nblocks(long long total, int bs) { return (unsigned)((total + bs - 1) / bs); }
inline int esize(int dt) { return dt == Y5_F16 ? 2 : dt == Y5_F32 ? 4 : dt == Y5_U8 ? 1 : 0; }
}  // namespace

extern "C" int y5_nchw_to_nhwc(const void* src, int sdt, void* dst, int ddt, int B, int C, int H, int W, int ld,
                               float scale, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!src || !dst || B <= 0 || C <= 0 || ld < C || ld > 64) return y5_fail(Y5_ERR_BAD_ARG, "nchw_to_nhwc: bad args (C <= ld <= 64)");
  const long long HW = (long long)H * W, npix = HW * B;
  const dim3 g(nblocks(npix, 256)), b(256);
#define Y5_CASE(S, D) hipLaunchKernelGGL((y5_nchw_to_nhwc_kernel<S, D>), g, b, 0, st, (const S*)src, (D*)dst, C, HW, npix, ld, scale)
  if (sdt == Y5_U8 && ddt == Y5_F16) Y5_CASE(unsigned char, half_t);
  else if (sdt == Y5_U8 && ddt == Y5_F32) Y5_CASE(unsigned char, float);
  else if (sdt == Y5_F16 && ddt == Y5_F16) Y5_CASE(half_t, half_t);
  else if (sdt == Y5_F16 && ddt == Y5_F32) Y5_CASE(half_t, float);
  else if (sdt == Y5_F32 && ddt == Y5_F16) Y5_CASE(float, half_t);
  else if (sdt == Y5_F32 && ddt == Y5_F32) Y5_CASE(float, float);
  else return y5_fail(Y5_ERR_BAD_ARG, "nchw_to_nhwc: unsupported dtype pair");
#undef Y5_CASE
  return y5_check_launch("y5_nchw_to_nhwc");
}

extern "C" int y5_nhwc_to_nchw(const void* src, int dt, void* dst, int B, int C, int H, int W, int ld, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!src || !dst || ld < C) return y5_fail(Y5_ERR_BAD_ARG, "nhwc_to_nchw: bad args");
  const long long HW = (long long)H * W, total = HW * B * C;
  const dim3 g(nblocks(total, 256)), b(256);
  if (dt == Y5_F16) hipLaunchKernelGGL((y5_nhwc_to_nchw_kernel<half_t>), g, b, 0, st, (const half_t*)src, (half_t*)dst, C, HW, total, ld);
  else if (dt == Y5_F32) hipLaunchKernelGGL((y5_nhwc_to_nchw_kernel<float>), g, b, 0, st, (const float*)src, (float*)dst, C, HW, total, ld);
  else return y5_fail(Y5_ERR_BAD_ARG, "nhwc_to_nchw: dtype");
  return y5_check_launch("y5_nhwc_to_nchw");
}

extern "C" int y5_sppf_pool(void* buf, int dt, int B, int H, int W, int C, int ld, int k, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int es = esize(dt);
  if (!buf || es < 2 || (C * es) % 16 || (ld * es) % 16 || ld < 4 * C || !(k & 1)) return y5_fail(Y5_ERR_BAD_ARG, "sppf_pool: bad args");
  // channel-group width, measured at the yolov5s shape (20x20, 256 of 1024 channels, bs 64) with the separable kernel and per-thread precomputed
  // windows (round 3): 16 B per pixel 46 us, 32 B 28 us, 64 B 23 us (whole 64-byte sectors of the 2 KiB-strided rows).  Y5_SPPF_GV overrides.
  int gv = 4;
  if (const char* e = getenv("Y5_SPPF_GV")) gv = atoi(e);
  if (gv != 1 && gv != 2 && gv != 4 && gv != 8) gv = 4;
  while (gv > 1 && ((C * es) % (16 * gv) != 0 || (size_t)H * W * 16 * gv * 3 > 150 * 1024)) gv >>= 1;
  const bool sep = (size_t)H * W * 16 * gv * 3 <= 150 * 1024;  // separable row / column passes need a third plane; otherwise the k x k window directly
  const size_t lds = (size_t)H * W * 16 * gv * (sep ? 3 : 2);
  if (lds > 150 * 1024) return y5_fail(Y5_ERR_UNSUPPORTED, "sppf_pool: H*W plane does not fit in LDS");
  const dim3 g((unsigned)(B * (C * es / (16 * gv)))), b(256);
#define Y5_SPPF_LAUNCH(V, G)                                                                                                             \
  do {                                                                                                                                   \
    static bool attr = false;                                                                                                            \
    if (!attr) {                                                                                                                         \
      hipFuncSetAttribute((const void*)y5_sppf_pool_kernel<V, G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);         \
      hipFuncSetAttribute((const void*)y5_sppf_pool_kernel<V, G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);        \
      attr = true;                                                                                                                       \
    }                                                                                                                                    \
    if (sep) hipLaunchKernelGGL((y5_sppf_pool_kernel<V, G, true>), g, b, lds, st, (char*)buf, H, W, C * es, ld * es, k);                 \
    else hipLaunchKernelGGL((y5_sppf_pool_kernel<V, G, false>), g, b, lds, st, (char*)buf, H, W, C * es, ld * es, k);                    \
  } while (0)
  if (dt == Y5_F16) {
    switch (gv) { case 8: Y5_SPPF_LAUNCH(half8_t, 8); break; case 4: Y5_SPPF_LAUNCH(half8_t, 4); break; case 2: Y5_SPPF_LAUNCH(half8_t, 2); break;
                  default: Y5_SPPF_LAUNCH(half8_t, 1); }
  } else {
    switch (gv) { case 8: Y5_SPPF_LAUNCH(float4_t, 8); break; case 4: Y5_SPPF_LAUNCH(float4_t, 4); break; case 2: Y5_SPPF_LAUNCH(float4_t, 2); break;
                  default: Y5_SPPF_LAUNCH(float4_t, 1); }
  }
#undef Y5_SPPF_LAUNCH
  return y5_check_launch("y5_sppf_pool");
}

extern "C" int y5_upsample2x(const void* src, int dt, void* dst, int B, int H, int W, int C, int lds, int ldd, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int es = esize(dt);
  if (!src || !dst || es < 2 || (C * es) % 16 || (lds * es) % 16 || (ldd * es) % 16) return y5_fail(Y5_ERR_BAD_ARG, "upsample2x: bad args");
  const int vpp = C * es / 16;
  const long long total = (long long)B * 4 * H * W * vpp;
  hipLaunchKernelGGL(y5_upsample2x_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, (const char*)src, (char*)dst, H, W, vpp,
                     lds * es, ldd * es, total);
  return y5_check_launch("y5_upsample2x");
}

extern "C" int y5_copy_slice(const void* src, int dt, void* dst, int npix, int C, int lds, int ldd, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int es = esize(dt);
  if (!src || !dst || es < 2 || (C * es) % 16 || (lds * es) % 16 || (ldd * es) % 16) return y5_fail(Y5_ERR_BAD_ARG, "copy_slice: bad args");
  const int vpp = C * es / 16;
  const long long total = (long long)npix * vpp;
  hipLaunchKernelGGL(y5_copy_slice_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, (const char*)src, (char*)dst, vpp, lds * es,
                     ldd * es, total);
  return y5_check_launch("y5_copy_slice");
}

extern "C" int y5_detect_decode_hint(const void* logits, int dt, int B, int ny, int nx, int na, int no, int nm, int ld, float stride,
                                     const float* anchors_px, void* z, int zdt, long long nrows_total, long long row_off, void* raw,
                                     void* obj_hint, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!logits || (!z && !raw) || !anchors_px || na < 1 || na > 8 || ld < na * no || nm < 0 || nm > no - 5)
    return y5_fail(Y5_ERR_BAD_ARG, "detect_decode: bad args");
  Y5DecodeParams p{};
  p.logits = logits; p.z = z; p.raw = raw; p.obj_hint = obj_hint;
  p.nrows_total = nrows_total; p.row_off = row_off;
  p.ny = ny; p.nx = nx; p.na = na; p.no = no; p.nm = nm; p.ld = ld; p.stride = stride;
  for (int i = 0; i < na * 2; ++i) p.anchors_px[i] = anchors_px[i];
  const int es = esize(dt);
  if ((ld * es) % 16 || ((uintptr_t)logits & 15)) return y5_fail(Y5_ERR_BAD_ARG, "detect_decode: logits rows must be 16-byte aligned");
  p.P = dt == Y5_F16 ? 64 : 32;
  while (p.P > 2 && (size_t)p.P * ld * es > 96 * 1024) p.P >>= 1;
  if ((long long)p.P * no >= 65536) return y5_fail(Y5_ERR_UNSUPPORTED, "detect_decode: no too large");
  p.inv_no = (unsigned)((0x100000000ULL + (unsigned)no - 1) / (unsigned)no);
  p.inv_nx = (unsigned)((0x100000000ULL + (unsigned)nx - 1) / (unsigned)nx);
  const size_t lds = (size_t)p.P * ld * es;
  const int npix = ny * nx;
  const dim3 g((unsigned)((npix + p.P - 1) / p.P), (unsigned)B), b(256);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)y5_detect_decode_kernel<half_t, half_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)y5_detect_decode_kernel<half_t, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)y5_detect_decode_kernel<float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr = true;
  }
  if (dt == Y5_F16 && zdt == Y5_F16) hipLaunchKernelGGL((y5_detect_decode_kernel<half_t, half_t>), g, b, lds, st, p);
  else if (dt == Y5_F16 && zdt == Y5_F32) hipLaunchKernelGGL((y5_detect_decode_kernel<half_t, float>), g, b, lds, st, p);
  else if (dt == Y5_F32 && zdt == Y5_F32) hipLaunchKernelGGL((y5_detect_decode_kernel<float, float>), g, b, lds, st, p);
  else return y5_fail(Y5_ERR_BAD_ARG, "detect_decode: dtype pair");
  return y5_check_launch("y5_detect_decode");
}

extern "C" int y5_detect_decode(const void* logits, int dt, int B, int ny, int nx, int na, int no, int nm, int ld, float stride,
                                const float* anchors_px, void* z, int zdt, long long nrows_total, long long row_off, void* raw,
                                void* stream_) {
  return y5_detect_decode_hint(logits, dt, B, ny, nx, na, no, nm, ld, stride, anchors_px, z, zdt, nrows_total, row_off, raw, nullptr, stream_);
}

// tiled fp16 head-layout kernels used by the training path (train_misc.hip declares the simple per-element versions)
extern "C" int y5_raw_to_nhwc_tiled(const void* draw, void* dlogits, int B, int npix, int na, int no, int ld, void* stream_) {
  if (!draw || !dlogits || B < 1 || npix < 1 || na < 1 || no < 1 || ld < na * no || (ld & 7) || (((uintptr_t)draw | (uintptr_t)dlogits) & 15))
    return y5_fail(Y5_ERR_BAD_ARG, "raw_to_nhwc: bad args");
  int P = 64;
  while (P > 2 && (size_t)P * ld * 2 > 96 * 1024) P >>= 1;
  if ((long long)P * no >= 65536) return y5_fail(Y5_ERR_UNSUPPORTED, "raw_to_nhwc: no too large");
  const unsigned inv_no = (unsigned)((0x100000000ULL + (unsigned)no - 1) / (unsigned)no);
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)y5_raw_to_nhwc_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
  hipLaunchKernelGGL(y5_raw_to_nhwc_tiled_kernel, dim3((unsigned)((npix + P - 1) / P), (unsigned)B), dim3(256), (size_t)P * ld * 2,
                     static_cast<hipStream_t>(stream_), (const half_t*)draw, (half_t*)dlogits, npix, na, no, ld, P, inv_no);
  return y5_check_launch("y5_raw_to_nhwc");
}
