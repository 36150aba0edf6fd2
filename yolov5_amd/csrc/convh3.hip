// Host-side launcher of the halo-resident 3x3 convolution (conv_h3.h); reached through y5_conv2d_fwd (conv.hip), configuration ids 61..
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "conv_h3.h"
#include "conv_pwk.h"
#include "y5_host.h"

namespace {

// ---- halo-resident 3x3 configurations (conv_h3.h): ids kH3_0 + index ------------------------------------------------------------
struct H3Cfg { int wm, wn, tm, tn, hpmax; };
constexpr int kNumH3 = 20;
constexpr H3Cfg kH3Cfgs[kNumH3] = {
    {2, 2, 5, 2, 496},  // 61: 320 pixels x 128 channels (8 x 40, 4 x 80, 16 x 20 output tiles)
    {2, 2, 5, 1, 496},  // 62: 320 x  64
    {2, 2, 7, 2, 512},  // 63: 448 x 128 (10 x 40: four tiles per 40 x 40 image; 20 x 20 whole images)
    {2, 2, 4, 2, 400},  // 64: 256 x 128 (5 x 40, 10 x 20)
    {2, 2, 7, 1, 512},  // 65: 448 x  64
    {2, 2, 4, 1, 400},  // 66: 256 x  64
    // eight waves (two per SIMD: one wave's LDS latency and DMA issue hide behind the other's MFMAs)
    {2, 4, 5, 1, 496},  // 67: 320 x 128
    {2, 4, 7, 1, 512},  // 68: 448 x 128
    {2, 4, 4, 1, 400},  // 69: 256 x 128
    {4, 2, 2, 2, 400},  // 70: 256 x 128, waves 4 x 2
    // 128 x 128 (four waves) / 128 x 64 (eight waves) register tiles per wave: 2 MFMAs per fragment read
    {4, 1, 4, 4, 512},  // 71: 512 x 128
    {4, 2, 4, 2, 512},  // 72: 512 x 128, eight waves
    // 4-stage filter ring: two workgroups per CU (independent barriers: one's LDS-DMA issue and epilogue overlap the other's MFMAs)
    {2, 2, 4, 2, 320},  // 73: 256 x 128 (5 x 40, 10 x 20)
    {2, 2, 4, 1, 320},  // 74: 256 x  64
    {2, 2, 3, 2, 320},  // 75: 192 x 128
    {4, 2, 2, 2, 320},  // 76: 256 x 128, eight waves, two workgroups per CU
    {4, 2, 2, 1, 320},  // 77: 256 x  64, eight waves, two workgroups per CU
    // round 5 (ids 90..92): small pixel tiles for the STRIDE-2 layers, whose halo is ~4.6x the output tile (4 x 16 outputs <- 9 x 33 inputs)
    {2, 2, 1, 2, 320},  // 90:  64 x 128, four waves, 4-stage ring: two workgroups per CU
    {4, 2, 1, 2, 592},  // 91: 128 x 128, eight waves, 4-stage ring (8 x 16 outputs <- 17 x 33 inputs)
    {2, 2, 2, 2, 592},  // 92: 128 x 128, four waves, 4-stage ring
};

// spatial tile for an H x W output: TW = ceil(W / d), TH as tall as the pixel budget and the LDS halo allow, then evened out over the
// image height; the candidate that needs the fewest rounds of `slots` concurrent workgroups wins, ties go to the smaller staged halo
bool h3_pick_tile(int B, int H, int W, int BM, int hpmax, long long tiles_n, long long slots, int S, int* th, int* tw) {
  long long best = -1;
  for (int d = 1; d <= 16 && d <= W; ++d) {
    const int TW = (W + d - 1) / d;
    if (TW > BM || (TW - 1) * S + 3 > 255) continue;
    int thm = BM / TW < H ? BM / TW : H;
    while (thm >= 1 && ((thm - 1) * S + 3) * ((TW - 1) * S + 3) > hpmax) --thm;
    if (thm < 1) continue;
    while ((thm - 1) * S + 3 > 255) --thm;
    const int nth = (H + thm - 1) / thm;
    const int TH = (H + nth - 1) / nth;
    const int ntw = (W + TW - 1) / TW;
    const long long tiles = (long long)B * nth * ntw * tiles_n;
    const long long rounds = (tiles + slots - 1) / slots;
    const long long cost = rounds * (1LL << 32) + (long long)((TH - 1) * S + 3) * ((TW - 1) * S + 3) * nth * ntw;
    if (best < 0 || cost < best) { best = cost; *th = TH; *tw = TW; }
  }
  return best >= 0;
}

template <int WM, int WN, int TM, int TN, int HPMAX, int NSW = 9>
int launch_h3(const Y5ConvParams& p0, int max_blocks, hipStream_t stream) {
  using Gm = Y5H3Geom<WM, WN, TM, TN, HPMAX, NSW>;
  static_assert(Gm::LDS <= 160 * 1024, "halo configuration exceeds 160 KiB of LDS");
  auto kern = y5_conv_h3_kernel<WM, WN, TM, TN, HPMAX, NSW>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  Y5ConvParams p = p0;
  if (p.Npad > Gm::BIAS_MAX) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: halo 3x3 configurations stage at most 1024 output channels of bias");
  p.tilesN = (p.Npad + Gm::BN - 1) / Gm::BN;
  long long G = max_blocks;
  if (G <= 0) {
    const int g_num_cu = y5_num_cu();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), Gm::NW * 64, Gm::LDS) != hipSuccess || occ < 1) occ = 1;
    G = (long long)g_num_cu * occ;
  }
  int th = 0, tw = 0;
  if (!h3_pick_tile(p.B, p.OH, p.OW, Gm::BM, HPMAX, p.tilesN, G, p.SH, &th, &tw))
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv: no spatial tile of this halo configuration fits the layer");
  p.h3_th = th; p.h3_tw = tw;
  p.h3_tiles_h = (p.OH + th - 1) / th;
  p.h3_tiles_w = (p.OW + tw - 1) / tw;
  const long long ntiles = (long long)p.B * p.h3_tiles_h * p.h3_tiles_w * p.tilesN;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return y5_fail(Y5_ERR_BAD_ARG, "conv: grid out of range");
  if (G > ntiles) G = ntiles;
  if (G >= 8) G &= ~7LL;
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(Gm::NW * 64), Gm::LDS, stream, p);
  return y5_check_launch("y5_conv2d_fwd(h3)");
}

}  // namespace

void y5_h3_cfg_info(int idx, int* bm, int* bn) {
  const H3Cfg& c = kH3Cfgs[idx < 0 || idx >= kNumH3 ? 0 : idx];
  *bm = c.wm * c.tm * 32;
  *bn = c.wn * c.tn * 32;
}

int y5_launch_h3_by_cfg(const Y5ConvParams& p, int idx, int mb, hipStream_t s) {
  switch (idx) {
    case 0: return launch_h3<2, 2, 5, 2, 496>(p, mb, s);
    case 1: return launch_h3<2, 2, 5, 1, 496>(p, mb, s);
    case 2: return launch_h3<2, 2, 7, 2, 512>(p, mb, s);
    case 3: return launch_h3<2, 2, 4, 2, 400>(p, mb, s);
    case 4: return launch_h3<2, 2, 7, 1, 512>(p, mb, s);
    case 5: return launch_h3<2, 2, 4, 1, 400>(p, mb, s);
    case 6: return launch_h3<2, 4, 5, 1, 496>(p, mb, s);
    case 7: return launch_h3<2, 4, 7, 1, 512>(p, mb, s);
    case 8: return launch_h3<2, 4, 4, 1, 400>(p, mb, s);
    case 9: return launch_h3<4, 2, 2, 2, 400>(p, mb, s);
    case 10: return launch_h3<4, 1, 4, 4, 512>(p, mb, s);
    case 11: return launch_h3<4, 2, 4, 2, 512>(p, mb, s);
    case 12: return launch_h3<2, 2, 4, 2, 320, 4>(p, mb, s);
    case 13: return launch_h3<2, 2, 4, 1, 320, 4>(p, mb, s);
    case 14: return launch_h3<2, 2, 3, 2, 320, 4>(p, mb, s);
    case 15: return launch_h3<4, 2, 2, 2, 320, 4>(p, mb, s);
    case 16: return launch_h3<4, 2, 2, 1, 320, 4>(p, mb, s);
    case 17: return launch_h3<2, 2, 1, 2, 320, 4>(p, mb, s);
    case 18: return launch_h3<4, 2, 1, 2, 592, 4>(p, mb, s);
    case 19: return launch_h3<2, 2, 2, 2, 592, 4>(p, mb, s);
  }
  return y5_fail(Y5_ERR_BAD_ARG, "conv: unknown halo 3x3 config");
}

// ---- K-streamed pointwise kernel (conv_pwk.h): ids 93 (256-channel N tile), 94 (128) -----------------------------------------------------------
template <int NT>
static int launch_pwk(const Y5ConvParams& p, hipStream_t stream) {
  using Gm = Y5PwkGeom<4, NT>;
  auto kern = y5_conv_pwk_kernel<4, NT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long gx = ((long long)p.M + Gm::BM - 1) / Gm::BM, gy = (p.Npad + Gm::BN - 1) / Gm::BN;
  if (gx < 1 || gx > 0x7fffffffLL || gy > 65535) return y5_fail(Y5_ERR_BAD_ARG, "conv: grid out of range");
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(Gm::NW * 64), Gm::LDS, stream, p);
  return y5_check_launch("y5_conv2d_fwd(pwk)");
}
int y5_launch_pwk_by_cfg(const Y5ConvParams& p, int idx, hipStream_t s) {
  switch (idx) {
    case 0: return launch_pwk<8>(p, s);
    case 1: return launch_pwk<4>(p, s);
  }
  return y5_fail(Y5_ERR_BAD_ARG, "conv: unknown K-streamed pointwise config");
}

#ifdef Y5_H3_TIMING
extern "C" int y5_h3_dbg_read(unsigned long long* dbg, unsigned long long* blocks) {  // kernel-experiment builds only (not part of the ABI)
  if (hipMemcpyFromSymbol(dbg, HIP_SYMBOL(y5_h3_dbg), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
  return hipMemcpyFromSymbol(blocks, HIP_SYMBOL(y5_h3_blocks), sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : -1;
}
#endif

// ---- y5_probe_mfma: what the matrix cores of this device sustain (bench.py reports it beside the data-sheet peak) ---------------------------
namespace {
__global__ __launch_bounds__(256) void y5_mfma_probe_kernel(float* out, unsigned long long* ticks, int iters) {
  float16_t acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (half_t)(threadIdx.x * 0.001f + e); b[e] = (half_t)(e * 0.5f - threadIdx.x * 0.002f); }
#ifndef Y5_EMU
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
#ifndef Y5_EMU
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = __builtin_amdgcn_s_memtime() - t0; ticks[1] = __builtin_amdgcn_s_memrealtime() - r0; }
#endif
}
}  // namespace

extern "C" int y5_probe_mfma(void* scratch, size_t scratch_bytes, int iters, float* tflops, float* shader_ghz, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  int dev = 0, ncu = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (ncu <= 0) ncu = 256;
  const int grid = 2 * ncu;
  const size_t need = (size_t)grid * 256 * 4 + 64;
  if (!scratch || scratch_bytes < need || iters < 1 || !tflops) return y5_fail(Y5_ERR_BAD_ARG, "probe_mfma: scratch too small / bad args");
  float* out = static_cast<float*>(scratch);
  unsigned long long* ticks = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch) + (size_t)grid * 256 * 4);
  hipLaunchKernelGGL(y5_mfma_probe_kernel, dim3(grid), dim3(256), 0, st, out, ticks, iters / 10 + 1);  // warm-up
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "probe_mfma: event create failed");
  hipEventRecord(e0, st);
  hipLaunchKernelGGL(y5_mfma_probe_kernel, dim3(grid), dim3(256), 0, st, out, ticks, iters);
  hipEventRecord(e1, st);
  int rc = Y5_OK;
  float ms = 0.f;
  if (hipEventSynchronize(e1) != hipSuccess) rc = y5_fail(Y5_ERR_RUNTIME, "probe_mfma: sync failed");
  else hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (rc) return rc;
  unsigned long long h[2] = {0, 0};
#ifdef Y5_EMU
  memcpy(h, ticks, sizeof(h));
#else
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
#endif
  *tflops = (float)((double)grid * 4 * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12);
  if (shader_ghz) *shader_ghz = h[1] ? (float)((double)h[0] / ((double)h[1] * 10.0)) : 0.f;
  return y5_check_launch("y5_probe_mfma");
}
