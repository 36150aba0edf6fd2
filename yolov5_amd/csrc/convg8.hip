// Host-side launcher of the 256-row / 8-phase implicit-GEMM family (conv_g8.h); reached through y5_conv2d_fwd (conv.hip), configuration ids 95 ..
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "conv_g8.h"
#include "y5_host.h"

// idx 0 (id 95): 256 pixels x 256 channels; idx 1 (id 96): 256 pixels x 128 channels; K tile 64, mfma_f32_32x32x16_f16
template <typename Gm, typename K>
static int launch_g8(K kern, const Y5ConvParams& p0, int max_blocks, hipStream_t stream, bool seq_kernel = false) {
  Y5ConvParams p = p0;
  p.tilesM = (p.M + Gm::BM - 1) / Gm::BM;
  p.tilesN = (p.Npad + Gm::BN - 1) / Gm::BN;
  p.nk = (p.K + Gm::BK - 1) / Gm::BK;   // (C1 % 64 != 0: the last K tile runs into the zero padding of the filter rows, Kpad % 64 == 0)
  y5_conv_set_fastdiv(p);
  // Stride-2 3x3: taps grouped by the class of input pixel they touch -- (odd row, odd column): the four corner taps; (odd, even): (0,1), (2,1); (even, odd):
  // (1,0), (1,2); (even, even): the centre -- so every re-request of a cache line follows within the next three K tiles (Y5ConvParams::tap_seq)
  p.tap_seq = 0;
  if (seq_kernel) {
    static const int seq[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
    for (int i = 0; i < 9; ++i) p.tap_seq |= (unsigned long long)seq[i] << (4 * i);
  }
  static const void* attr_done[4] = {nullptr, nullptr, nullptr, nullptr};   // (the instantiations of one geometry share this function: same pointer type)
  bool seen = false;
  for (const void* a : attr_done) seen = seen || a == reinterpret_cast<const void*>(kern);
  if (!seen) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (const void*& a : attr_done)
      if (!a) { a = reinterpret_cast<const void*>(kern); break; }
  }
  const long long ntiles = (long long)p.tilesM * p.tilesN;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return y5_fail(Y5_ERR_BAD_ARG, "conv: grid out of range");
  long long G = max_blocks > 0 ? max_blocks : y5_num_cu();   // > 80 KiB of LDS: one workgroup per CU
  if (G >= ntiles) G = ntiles;        // one tile each: no constraint from the XCD remap
  else if (G >= 8) G &= ~7LL;         // several tiles per workgroup: bid % 8 must stay the XCD of every virtual id bid + j * G
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(Gm::NW * 64), Gm::LDS, stream, p);
  return y5_check_launch("y5_conv2d_fwd(g8)");
}

int y5_launch_g8_by_cfg(const Y5ConvParams& p, int idx, int max_blocks, hipStream_t stream) {
  const bool up = p.up_c > 0;   // virtual Upsample + Concat loader (1x1 s1 layers; validated by the caller)
  const bool gen = (p.C1 % 64) != 0;   // general-C1 loader (conv_g8.h GEN): a K tile may span two taps
  if (p.C1 % 8 || p.C1 < 64 || p.KH * p.KW > (gen ? 31 : 32) || p.Kpad % 64 || p.Kpad < p.K || p.Npad > Y5G8Geom::MAXN || (gen && up))
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv: the 8-phase configurations need C1 % 8 == 0, C1 >= 64 (C1 % 64 == 0 with up_c > 0), Kpad % 64 == 0, KH * KW <= 32 (31 unless C1 % 64 == 0) and Npad <= 2048");
  const bool seq = !up && !gen && p.KH == 3 && p.KW == 3 && p.SH == 2 && p.SW == 2;   // class-ordered taps (launch_g8 fills Y5ConvParams::tap_seq)
  switch (idx) {
    case 0:
      if (gen) return launch_g8<Y5G8Geom>(y5_conv_g8_kernel<false, false, true>, p, max_blocks, stream);
      if (seq) return launch_g8<Y5G8Geom>(y5_conv_g8_kernel<false, true>, p, max_blocks, stream, true);
      return up ? launch_g8<Y5G8Geom>(y5_conv_g8_kernel<true>, p, max_blocks, stream) : launch_g8<Y5G8Geom>(y5_conv_g8_kernel<false>, p, max_blocks, stream);
    case 1:
      if (gen) return launch_g8<Y5G8nGeom>(y5_conv_g8n_kernel<false, false, true>, p, max_blocks, stream);
      if (seq) return launch_g8<Y5G8nGeom>(y5_conv_g8n_kernel<false, true>, p, max_blocks, stream, true);
      return up ? launch_g8<Y5G8nGeom>(y5_conv_g8n_kernel<true>, p, max_blocks, stream) : launch_g8<Y5G8nGeom>(y5_conv_g8n_kernel<false>, p, max_blocks, stream);
  }
  return y5_fail(Y5_ERR_BAD_ARG, "conv: unknown 8-phase config");
}

#ifdef Y5_G8_TIMING
extern "C" int y5_g8_dbg_read(unsigned long long* out) {  // kernel-experiment builds only (not part of the ABI)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(y5_g8_dbg), sizeof(unsigned long long) * 512 * 2 * 8) == hipSuccess ? 0 : -1;
}
#endif
