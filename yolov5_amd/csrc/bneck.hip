// Fused Bottleneck (conv_bneck.h): y = [x +] SiLU(W2 (*) SiLU(W1 x + b1) + b2) with the 1x1 output kept in LDS -- models/common.py:164-181
// inside C3 (e = 1.0, :242) at the HBM-bound high-resolution levels (C = 32 at P2, 64 at P3 of yolov5s).
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "conv_bneck.h"
#include "y5_host.h"

namespace {
template <int C, int S, bool ADD>
int launch_bneck(const Y5BneckParams& p, int max_blocks, hipStream_t stream) {
  const size_t lds = y5_conv_bneck_lds_bytes<C, S>();
  if (lds > 160 * 1024) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: exceeds 160 KiB of LDS");
  auto kern = y5_conv_bneck_kernel<C, S, ADD>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long nwt = (long long)p.B * (p.H / 4) * (p.W / 8);
  const long long nbt = (nwt + 3) >> 2;
  long long G = max_blocks;
  if (G <= 0) {
    static int num_cu = 0;
    if (!num_cu) {
      int dev = 0, n = 0;
      hipGetDevice(&dev);
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      num_cu = n > 0 ? n : 256;
    }
    // resident workgroups per CU: LDS is the limit (52.5 KB at C = 32, one stage: three fit in 160 KB; the occupancy query answers two)
    int occ = (int)((160 * 1024) / lds);
    if (occ < 1) occ = 1;
    if (occ > 4) occ = 4;
    G = (long long)num_cu * occ;
  }
  if (G > nbt) G = nbt;
  if (G >= 8) G &= ~7LL;
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(256), lds, stream, p);
  return y5_check_launch("y5_bottleneck_fwd");
}
}  // namespace

extern "C" int y5_bottleneck_fwd(const void* x, int ldx, const void* w1_packed, const float* bias1, int Kpad1, const void* w2_packed,
                                 const float* bias2, int Kpad2, void* y, int ldy, int B, int H, int W, int C, int add, int max_blocks,
                                 void* stream_) {
  if (!x || !w1_packed || !bias1 || !w2_packed || !bias2 || !y) return y5_fail(Y5_ERR_BAD_ARG, "bottleneck: null pointer");
  if (C != 32 && C != 64) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: fused kernel exists for 32 and 64 channels");
  if (B < 1 || H < 4 || W < 8 || (H & 3) || (W & 7) || H > 255 * 4 || W > 65535) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: needs H % 4 == 0 and W % 8 == 0");
  if ((ldx & 7) || (ldy & 7) || ldx < C || ldy < C || Kpad1 < C || (Kpad1 & 7) || Kpad2 < 9 * C || (Kpad2 & 7))
    return y5_fail(Y5_ERR_BAD_ARG, "bottleneck: bad strides / packed filter dims");
  if (((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)bias1 | (uintptr_t)w2_packed | (uintptr_t)bias2 | (uintptr_t)y) & 15)
    return y5_fail(Y5_ERR_BAD_ARG, "bottleneck: pointers must be 16-byte aligned");
  const long long npix = (long long)B * H * W;
  if (npix * ldx * 2 >= 0x7fffffffLL || npix * ldy * 2 >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: tensor exceeds 2^31 bytes");
  const char* xb = static_cast<const char*>(x);
  char* yb = static_cast<char*>(y);
  if (xb < yb + npix * ldy * 2 && yb < xb + npix * ldx * 2 && (ldx != ldy || ((yb - xb) % (ldx * 2) + ldx * 2) % (ldx * 2) < C * 2 || ((xb - yb) % (ldx * 2) + ldx * 2) % (ldx * 2) < C * 2))
    return y5_fail(Y5_ERR_BAD_ARG, "bottleneck: y must not overlap x (a tile reads its neighbours' pixels of x)");
  Y5BneckParams p{};
  p.x = x; p.w1 = w1_packed; p.w2 = w2_packed; p.b1 = bias1; p.b2 = bias2; p.y = y;
  p.B = B; p.H = H; p.W = W; p.ldx = ldx; p.ldy = ldy; p.Kpad1 = Kpad1; p.Kpad2 = Kpad2; p.add = add;
  p.x_bytes = (unsigned)(((npix - 1) * ldx + C) * 2);
  p.w1_bytes = (unsigned)((long long)C * Kpad1 * 2);
  p.w2_bytes = (unsigned)((long long)C * Kpad2 * 2);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int stages = max_blocks >> 16;      // bits 16.. of max_blocks select the ring depth (0 = default for C), bits 0..15 the grid cap
  max_blocks &= 0xffff;
  const int S = stages > 0 ? stages : 1;  // measured (scripts/bneck_bench.py): resident waves beat prefetch depth -- 1 stage, 3 workgroups per CU
  if (C == 32) {
    if (S == 1) return add ? launch_bneck<32, 1, true>(p, max_blocks, st) : launch_bneck<32, 1, false>(p, max_blocks, st);
    if (S == 2) return add ? launch_bneck<32, 2, true>(p, max_blocks, st) : launch_bneck<32, 2, false>(p, max_blocks, st);
    if (S == 3) return add ? launch_bneck<32, 3, true>(p, max_blocks, st) : launch_bneck<32, 3, false>(p, max_blocks, st);
  } else if (S == 1) {
    return add ? launch_bneck<64, 1, true>(p, max_blocks, st) : launch_bneck<64, 1, false>(p, max_blocks, st);
  }
  return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: unsupported number of stages");
}
