// Host-side launcher + C-ABI entry point for the implicit-GEMM convolution (see conv_igemm.h).
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "conv_igemm.h"
#include "y5_host.h"

namespace {

template <typename T, int WM, int WN, int TM, int TN, bool TABLE>
int launch_cfg(const Y5ConvParams& p0, hipStream_t stream) {
  using Tr = Y5Tr<T>;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  Y5ConvParams p = p0;
  p.tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.Npad + BN - 1) / BN;
  p.nk = p.Kpad / Tr::BK;
  size_t lds = 2 * (BM + BN) * Y5_CONV_ROWB;
  if (TABLE) {
    if (p.Kpad / Tr::EPP > Y5_CONV_MAXTAB) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: K too large for gather-table mode");
    lds += (size_t)(p.Kpad / Tr::EPP) * 8;
  }
  auto kern = y5_conv_igemm_kernel<T, WM, WN, TM, TN, TABLE>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_done = true;
  }
  const long long nblk = (long long)p.tilesM * p.tilesN;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return y5_fail(Y5_ERR_BAD_ARG, "conv: grid out of range");
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(WM * WN * 64), lds, stream, p);
  return y5_check_launch("y5_conv2d_fwd");
}

template <typename T, bool TABLE>
int launch_by_n(const Y5ConvParams& p, int tile_n, hipStream_t stream) {
  switch (tile_n) {
    case 32: return launch_cfg<T, 4, 1, 1, 1, TABLE>(p, stream);
    case 64: return launch_cfg<T, 4, 1, 1, 2, TABLE>(p, stream);
    case 128: return launch_cfg<T, 2, 2, 2, 2, TABLE>(p, stream);
    case 256: return launch_cfg<T, 2, 2, 2, 4, TABLE>(p, stream);
  }
  return y5_fail(Y5_ERR_BAD_ARG, "conv: tile_n must be 32/64/128/256");
}

}  // namespace

extern "C" int y5_conv2d_fwd(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                             const void* residual, void* y, void* y_up2, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!d || !x || !w_packed || !bias || !y) return y5_fail(Y5_ERR_BAD_ARG, "conv: null pointer");
  const int es = d->dtype == Y5_F16 ? 2 : d->dtype == Y5_F32 ? 4 : 0;
  if (!es) return y5_fail(Y5_ERR_BAD_ARG, "conv: dtype must be Y5_F16 or Y5_F32");
  const int epp = 16 / es, bk = 64 / es;
  if (d->C1 % epp || d->ldx % epp) return y5_fail(Y5_ERR_BAD_ARG, "conv: C1 and ldx must be multiples of 16 bytes");
  if (d->C2 % 4 || d->ldy % 4 || (residual && d->ldr % 4) || (y_up2 && d->ld2 % 4))
    return y5_fail(Y5_ERR_BAD_ARG, "conv: C2/ldy/ldr/ld2 must be multiples of 4");
  if (d->Kpad % bk || d->Kpad < d->KH * d->KW * d->C1 || d->Npad % 32 || d->Npad < d->C2)
    return y5_fail(Y5_ERR_BAD_ARG, "conv: bad packed filter dims (Kpad % BK, Npad % 32)");
  if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)y_up2) & 15)
    return y5_fail(Y5_ERR_BAD_ARG, "conv: pointers must be 16-byte aligned");
  const int oh = (d->H + 2 * d->PH - d->KH) / d->SH + 1, ow = (d->W + 2 * d->PW - d->KW) / d->SW + 1;
  if (oh != d->OH || ow != d->OW) return y5_fail(Y5_ERR_BAD_ARG, "conv: OH/OW inconsistent with H/W/k/s/p");
  if ((long long)d->B * d->H * d->W * d->ldx >= 0x7fffffffLL || (long long)d->B * oh * ow >= 0x7fffffffLL)
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv: tensor exceeds 2^31 elements");

  Y5ConvParams p{};
  p.x = x; p.w = w_packed; p.bias = bias; p.res = residual; p.y = y; p.y2 = y_up2;
  p.zero = y5_zero_page();
  if (!p.zero) return y5_fail(Y5_ERR_RUNTIME, "conv: zero page allocation failed");
  p.B = d->B; p.H = d->H; p.W = d->W; p.C1 = d->C1; p.ldx = d->ldx;
  p.OH = oh; p.OW = ow; p.C2 = d->C2; p.ldy = d->ldy;
  p.KH = d->KH; p.KW = d->KW; p.SH = d->SH; p.SW = d->SW; p.PH = d->PH; p.PW = d->PW;
  p.act = d->act; p.Kpad = d->Kpad; p.Npad = d->Npad; p.K = d->KH * d->KW * d->C1;
  p.ldr = d->ldr; p.ld2 = d->ld2;
  p.M = d->B * oh * ow;

  int tile_n = d->tile_n;
  if (tile_n == 0) tile_n = d->Npad <= 32 ? 32 : d->Npad <= 64 ? 64 : 128;
  const bool table = (d->C1 % bk) != 0;
  if (d->dtype == Y5_F16)
    return table ? launch_by_n<half_t, true>(p, tile_n, stream) : launch_by_n<half_t, false>(p, tile_n, stream);
  return table ? launch_by_n<float, true>(p, tile_n, stream) : launch_by_n<float, false>(p, tile_n, stream);
}
