// Fused Detect head for the export / z-only path (models/yolo.py:83-108 with Detect.export, models/common.py:866): the 1x1 Detect
// convolution of one pyramid level (128 -> 3 x 85 channels) and the Detect decode in ONE pass of the streaming pointwise kernel
// (conv_pw.h, configuration 56: filter resident in LDS, per-wave rings) -- the logits never reach HBM (0.42 GB less traffic at
// P3 of yolov5s bs=64 640^2).  Results are bit-identical to y5_conv2d_fwd(act = 0) + y5_detect_decode(raw = NULL): same
// fp16-rounded logits, same decode arithmetic; this file is built with -ffp-contract=off like the decode kernel's.
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "conv_pw.h"
#include "conv_headk.h"
#include "y5_host.h"

extern "C" int y5_detect_head_fwd_hint(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, int ny, int nx, float stride,
                                       const float* anchors_px, void* z, long long nrows_total, long long row_off, void* obj_hint, void* stream_) {
  if (!d || !x || !w_packed || !bias || !anchors_px || !z) return y5_fail(Y5_ERR_BAD_ARG, "detect_head: null pointer");
  constexpr int KC = 2, RB = 128, NT = 8, S = 2, OS = 2;
  const long long npix = (long long)ny * nx;
  if (d->C1 > KC * RB / 2) {   // deep levels (P4 / P5: 256 / 512 input channels): the K-streamed kernel of conv_headk.h
    using Gm = Y5HeadkGeom<4>;
    if (d->dtype != Y5_F16 || d->KH != 1 || d->KW != 1 || d->SH != 1 || d->SW != 1 || d->PH || d->PW || d->act || (d->C1 & 31) || d->Npad != Gm::NPAD || d->C2 < 255 ||
        d->C2 > 256 || d->Kpad < d->C1 || (d->Kpad & 7) || d->H != ny || d->W != nx || d->OH != ny || d->OW != nx || d->ldx % 8 || d->out_mul_h != 0)
      return y5_fail(Y5_ERR_UNSUPPORTED, "detect_head: needs a 1x1 fp16 convolution C1 (multiple of 32) -> 3 x 85 channels without activation");
    const long long M = (long long)d->B * npix;
    if ((M & 31) || (npix & 7) || npix < 32 || npix >= 65536 || (nrows_total & 7) || (row_off & 7) || nrows_total < row_off + 3 * npix)
      return y5_fail(Y5_ERR_UNSUPPORTED, "detect_head: B * pixels must be a multiple of 32, pixels per image a multiple of 8 (32 .. 65535), z rows 8-row aligned");
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)z | (uintptr_t)obj_hint) & 15) return y5_fail(Y5_ERR_BAD_ARG, "detect_head: pointers must be 16-byte aligned");
    if (M * d->ldx * 2 >= 0x7fffffffLL || M >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "detect_head: tensor exceeds 2^31 bytes");
    Y5ConvParams p{};
    p.x = x; p.w = w_packed; p.bias = bias;
    p.B = d->B; p.H = ny; p.W = nx; p.C1 = d->C1; p.ldx = d->ldx; p.C2 = d->C2; p.Kpad = d->Kpad; p.Npad = d->Npad; p.M = (int)M;
    p.x_bytes = (unsigned)(((M - 1) * d->ldx + d->C1) * 2);
    p.w_bytes = (unsigned)((long long)d->Npad * d->Kpad * 2);
    Y5HeadParams h{};
    h.z = z; h.nrows_total = nrows_total; h.row_off = row_off; h.npix = (int)npix; h.nx = nx;
    h.inv_nx = (unsigned)((0x100000000ULL + (unsigned)nx - 1) / (unsigned)nx);
    h.stride = stride;
    for (int i = 0; i < 6; ++i) h.anchors_px[i] = anchors_px[i];
    h.obj_hint = obj_hint;
    static bool attr_k = false;
    if (!attr_k) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(y5_conv_headk_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(y5_conv_headk_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_k = true;
    }
    const unsigned G = (unsigned)((M + Gm::BM - 1) / Gm::BM);
    if (obj_hint) hipLaunchKernelGGL((y5_conv_headk_kernel<4, true>), dim3(G), dim3(Gm::NW * 64), Gm::LDS, static_cast<hipStream_t>(stream_), p, h);
    else hipLaunchKernelGGL((y5_conv_headk_kernel<4, false>), dim3(G), dim3(Gm::NW * 64), Gm::LDS, static_cast<hipStream_t>(stream_), p, h);
    return y5_check_launch("y5_detect_head_fwd(headk)");
  }
  if (d->dtype != Y5_F16 || d->KH != 1 || d->KW != 1 || d->SH != 1 || d->SW != 1 || d->PH || d->PW || d->act || d->C1 != KC * RB / 2 ||
      d->Npad != NT * 32 || d->C2 < 255 || d->C2 > 256 || d->Kpad * 2 < KC * RB || d->H != ny || d->W != nx || d->OH != ny || d->OW != nx ||
      d->ldx % 8 || d->out_mul_h != 0)
    return y5_fail(Y5_ERR_UNSUPPORTED, "detect_head: needs a 1x1 fp16 convolution 128 -> 3 x 85 channels without activation");
  if (npix % 32 || npix >= 65536 || (nrows_total & 7) || (row_off & 7) || (npix & 7) || nrows_total < row_off + 3 * npix)
    return y5_fail(Y5_ERR_UNSUPPORTED, "detect_head: pixels per image must be a multiple of 32 (< 65536), z rows 8-row aligned");
  if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)z) & 15) return y5_fail(Y5_ERR_BAD_ARG, "detect_head: pointers must be 16-byte aligned");
  if ((long long)d->B * npix * d->ldx * 2 >= 0x7fffffffLL || (long long)d->B * npix >= 0x7fffffffLL)
    return y5_fail(Y5_ERR_UNSUPPORTED, "detect_head: tensor exceeds 2^31 bytes");
  Y5ConvParams p{};
  p.x = x; p.w = w_packed; p.bias = bias;
  p.zero = y5_zero_page();
  if (!p.zero) return y5_fail(Y5_ERR_RUNTIME, "detect_head: zero page allocation failed");
  p.B = d->B; p.H = ny; p.W = nx; p.C1 = d->C1; p.ldx = d->ldx; p.OH = ny; p.OW = nx; p.C2 = d->C2; p.ldy = d->Npad;
  p.KH = p.KW = p.SH = p.SW = 1; p.Kpad = d->Kpad; p.Npad = d->Npad; p.K = d->C1;
  p.M = (int)(d->B * npix);
  p.x_bytes = (unsigned)((((long long)d->B * npix - 1) * d->ldx + d->C1) * 2);
  p.w_bytes = (unsigned)((long long)d->Npad * d->Kpad * 2);
  Y5HeadParams h{};
  h.z = z; h.nrows_total = nrows_total; h.row_off = row_off; h.npix = (int)npix; h.nx = nx;
  h.inv_nx = (unsigned)((0x100000000ULL + (unsigned)nx - 1) / (unsigned)nx);
  h.stride = stride;
  for (int i = 0; i < 6; ++i) h.anchors_px[i] = anchors_px[i];
  h.obj_hint = obj_hint;
  if (obj_hint && ((uintptr_t)obj_hint & 15)) return y5_fail(Y5_ERR_BAD_ARG, "detect_head: hint plane must be 16-byte aligned");

  // d->cfg == 87: eight waves per workgroup with one stage each (two waves per SIMD under one filter copy); otherwise four waves, two stages
  const bool w8 = d->cfg == 87;
  const int nwv = w8 ? 8 : 4;
  const size_t lds = w8 ? y5_conv_pw_lds_bytes<KC, RB, NT, 1, OS, 8>() : y5_conv_pw_lds_bytes<KC, RB, NT, S, OS>();
  auto kern = w8 ? (obj_hint ? y5_conv_pw_head_kernel<KC, RB, NT, 1, OS, true, 8> : y5_conv_pw_head_kernel<KC, RB, NT, 1, OS, false, 8>)
                 : (obj_hint ? y5_conv_pw_head_kernel<KC, RB, NT, S, OS, true> : y5_conv_pw_head_kernel<KC, RB, NT, S, OS, false>);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(y5_conv_pw_head_kernel<KC, RB, NT, S, OS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(y5_conv_pw_head_kernel<KC, RB, NT, S, OS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(y5_conv_pw_head_kernel<KC, RB, NT, 1, OS, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(y5_conv_pw_head_kernel<KC, RB, NT, 1, OS, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long nbt = ((long long)(p.M >> 5) + nwv - 1) / nwv;
  long long G = d->max_blocks;
  if (G <= 0) {
    const int num_cu = y5_num_cu();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), nwv * 64, lds) != hipSuccess || occ < 1) occ = 1;
    G = (long long)num_cu * occ;
  }
  if (G > nbt) G = nbt;
  if (G >= 8) G &= ~7LL;
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(nwv * 64), lds, static_cast<hipStream_t>(stream_), p, h);
  return y5_check_launch("y5_detect_head_fwd");
}

extern "C" int y5_detect_head_fwd(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, int ny, int nx, float stride,
                                  const float* anchors_px, void* z, long long nrows_total, long long row_off, void* stream_) {
  return y5_detect_head_fwd_hint(d, x, w_packed, bias, ny, nx, stride, anchors_px, z, nrows_total, row_off, nullptr, stream_);
}
