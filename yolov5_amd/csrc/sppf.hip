// SPPF front half in one launch (conv_sppf.h): cv1 (1x1 + bias + SiLU) and the three cascaded max pools of models/common.py:318-340, written to the
// four channel slices of the concat buffer cv2 reads.
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "conv_sppf.h"
#include "y5_host.h"

extern "C" int y5_sppf_cv1_pool_fwd(const void* x, int ldx, const void* w_packed, const float* bias, int Kpad, void* buf, int ld, int B, int H, int W,
                                    int C1, int c_, int k, int act, void* stream_) {
  using Gm = Y5SppfGeom<4>;
  if (!x || !w_packed || !bias || !buf) return y5_fail(Y5_ERR_BAD_ARG, "sppf_cv1_pool: null pointer");
  if (B < 1 || H < 1 || W < 1 || H * W > Gm::MAXHW || C1 < 32 || (C1 & 31) || c_ < 64 || (c_ & 63) || !(k & 1) || k < 1)
    return y5_fail(Y5_ERR_UNSUPPORTED, "sppf_cv1_pool: needs H * W <= 416 pixels per image, C1 % 32 == 0, c_ % 64 == 0, odd k");
  if ((ldx & 7) || (ld & 7) || ldx < C1 || ld < 4 * c_ || Kpad < C1 || (Kpad & 7)) return y5_fail(Y5_ERR_BAD_ARG, "sppf_cv1_pool: bad strides / packed filter dims");
  if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)buf) & 15) return y5_fail(Y5_ERR_BAD_ARG, "sppf_cv1_pool: pointers must be 16-byte aligned");
  const long long npix = (long long)B * H * W;
  if (npix * ldx * 2 >= 0x7fffffffLL || npix * ld * 2 >= 0x7fffffffLL || (long long)c_ * Kpad * 2 >= 0x7fffffffLL)
    return y5_fail(Y5_ERR_UNSUPPORTED, "sppf_cv1_pool: tensor exceeds 2^31 bytes");
  Y5SppfParams p{};
  p.x = x; p.w = w_packed; p.bias = bias; p.buf = buf;
  p.x_bytes = (unsigned)(((npix - 1) * ldx + C1) * 2);
  p.w_bytes = (unsigned)((long long)c_ * Kpad * 2);
  p.B = B; p.H = H; p.W = W; p.C1 = C1; p.ldx = ldx; p.c_ = c_; p.ld = ld; p.Kpad = Kpad; p.k = k; p.act = act;
  auto kern = y5_sppf_cv1_pool_kernel<4>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long G = (long long)B * (c_ / 64);
  if (G > 0x7fffffffLL) return y5_fail(Y5_ERR_BAD_ARG, "sppf_cv1_pool: grid out of range");
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(Gm::NW * 64), Gm::LDS, static_cast<hipStream_t>(stream_), p);
  return y5_check_launch("y5_sppf_cv1_pool_fwd");
}
