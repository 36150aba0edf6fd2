// C-ABI of the fused backbone front (csrc/conv_front.h): 0.Conv (k6 s2, from NCHW) + 1.Conv (k3 s2) + the pointwise layer behind it, one launch.
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "conv_front.h"
#include "y5_host.h"

namespace {

template <int TH, int TW, int NT1, int NT2>
int launch_front(const Y5FrontParams& p, int max_blocks, hipStream_t stream) {
  const size_t lds = y5_conv_front_lds_bytes<TH, TW, NT1, NT2>();
  auto kern = y5_conv_front_kernel<TH, TW, NT1, NT2>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int g_front_cu = y5_num_cu();
  const long long ntiles = (long long)p.B * p.tiles_h * p.tiles_w;
  long long G = max_blocks > 0 ? max_blocks : g_front_cu;  // one workgroup (eight waves, ~153 KB of LDS) per CU
  if (G > ntiles) G = ntiles;
  if (G >= 8) G &= ~7LL;
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(512), lds, stream, p);
  return y5_check_launch("y5_conv_front_fwd");
}
}  // namespace

extern "C" int y5_conv_front_fwd(const void* x_nchw, int B, int H, int W, const void* w_stem, const float* bias0, int C0, const void* w1_packed,
                                 const float* bias1, int C1, int Npad1, int Kpad1, int act1, const void* w2_packed, const float* bias2, int C3, int Npad2,
                                 int Kpad2, int act2, void* y, int ldy, void* y2, int ld2, int split_n, int max_blocks, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x_nchw || !w_stem || !bias0 || !w1_packed || !bias1 || !w2_packed || !bias2 || !y) return y5_fail(Y5_ERR_BAD_ARG, "conv_front: null pointer");
  if (B < 1 || H < 64 || W < 64 || (H & 63) || (W & 63)) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_front: needs H % 64 == 0 and W % 64 == 0 (16 x 16 tiles of the 3x3's output)");
  if (C0 != 32) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_front: built for a 32-channel stem (yolov5s / yolov5s-seg)");
  if (C1 < 8 || (C1 & 7) || C1 > Npad1 || (Npad1 != 32 && Npad1 != 64) || Kpad1 < 288 || (Kpad1 & 7))
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv_front: the 3x3 layer must have <= 64 output channels in multiples of 8");
  if (C3 < 16 || (C3 & 15) || C3 > Npad2 || (Npad2 != 32 && Npad2 != 64) || Kpad2 < Npad1 || (Kpad2 & 7) || split_n < 0 || split_n > C3 || (split_n & 7))
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv_front: the pointwise layer must have <= 64 output channels in multiples of 16, split on a multiple of 8");
  if (split_n < C3 && !y2) return y5_fail(Y5_ERR_BAD_ARG, "conv_front: split output needs y2");
  if ((ldy & 7) || (y2 && (ld2 & 7))) return y5_fail(Y5_ERR_BAD_ARG, "conv_front: pixel strides must be multiples of 8");
  if (((uintptr_t)x_nchw | (uintptr_t)w_stem | (uintptr_t)w1_packed | (uintptr_t)w2_packed | (uintptr_t)y | (uintptr_t)y2) & 15)
    return y5_fail(Y5_ERR_BAD_ARG, "conv_front: pointers must be 16-byte aligned");
  if ((long long)B * 3 * H * W * 2 >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_front: input exceeds 2^31 bytes");
  Y5FrontParams p{};
  p.x = x_nchw; p.w0 = w_stem; p.b0 = bias0; p.w1 = w1_packed; p.b1 = bias1; p.w2 = w2_packed; p.b2 = bias2; p.y = y; p.y2 = y2 ? y2 : y;
  p.x_bytes = (unsigned)((long long)B * 3 * H * W * 2);
  p.w1_bytes = (unsigned)((long long)Npad1 * Kpad1 * 2);
  p.w2_bytes = (unsigned)((long long)Npad2 * Kpad2 * 2);
  p.B = B; p.H = H; p.W = W; p.OH0 = H / 2; p.OW0 = W / 2; p.OH1 = H / 4; p.OW1 = W / 4;
  p.Kpad1 = Kpad1; p.Kpad2 = Kpad2; p.C3 = C3; p.split = split_n; p.ldy = ldy; p.ld2 = y2 ? ld2 : ldy;
  p.act1 = act1; p.act2 = act2;
  p.tiles_h = p.OH1 / 16; p.tiles_w = p.OW1 / 16;
  if (Npad1 == 64 && Npad2 == 64) return launch_front<16, 16, 2, 2>(p, max_blocks, stream);
  if (Npad1 == 32 && Npad2 == 32) return launch_front<16, 16, 1, 1>(p, max_blocks, stream);
  if (Npad1 == 64 && Npad2 == 32) return launch_front<16, 16, 2, 1>(p, max_blocks, stream);
  return launch_front<16, 16, 1, 2>(p, max_blocks, stream);
}

#ifdef Y5_FRONT_TIMING
extern "C" int y5_front_dbg_read(unsigned long long* out) {  // kernel-experiment builds only (not part of the ABI)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(y5_front_dbg), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif
