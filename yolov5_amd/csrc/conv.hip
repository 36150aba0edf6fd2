// Host-side launcher + C-ABI entry points for the implicit-GEMM convolution (see conv_igemm.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/yolov5_hip.h"
#include "conv_igemm.h"
#include "conv_pw.h"
#include "conv_stem.h"
#include "conv_k3.h"
#include "y5_host.h"

namespace {

struct TileCfg { int wm, wn, tm, tn, rb; };
// id -> workgroup tile (BM = wm*tm*32 pixels, BN = wn*tn*32 channels), LDS row bytes (K per stage = rb / elemsize)
constexpr int kNumIgemm = 14;
constexpr int kRing0 = 22, kNumRing = 8;  // ids 22..29: tile shapes of ids {1,2,3,5,7,8,11,12} with a 3-stage ring
constexpr int kRingBase[kNumRing] = {1, 2, 3, 5, 7, 8, 11, 12};
// ids 35..39: large tiles for the deep layers (K >= 576, N >= 128).  The general mainloop is bound by the L2->LDS bytes in
// flight per CU, so these trade occupancy for bytes per flop: 256-row tiles, BK32 chunks, deeper rings, 8 waves where the
// tile is 256 wide.  (wm, wn, tm, tn, rb, stages)
struct BigCfg { int wm, wn, tm, tn, rb, ns; };
constexpr int kBig0 = 35, kNumBig = 21;
constexpr BigCfg kBigCfgs[kNumBig] = {
    {2, 2, 4, 2, 64, 4},   // 35: 256 x 128, BK32, 4 stages, 4 waves
    {4, 2, 2, 2, 64, 4},   // 36: 256 x 128, BK32, 4 stages, 8 waves
    {2, 4, 4, 2, 64, 4},   // 37: 256 x 256, BK32, 4 stages, 8 waves
    {2, 4, 4, 2, 64, 3},   // 38: 256 x 256, BK32, 3 stages, 8 waves
    {2, 4, 4, 2, 128, 2},  // 39: 256 x 256, BK64, 2 stages, 8 waves
    // producer / consumer (conv_igemm.h PROD): as many LDS-DMA waves again as MFMA waves
    {2, 2, 4, 2, 128, 3},  // 40: 256 x 128, BK64, 3 stages, 4 + 4 waves
    {2, 2, 4, 2, 64, 4},   // 41: 256 x 128, BK32, 4 stages, 4 + 4 waves
    {2, 2, 2, 2, 128, 3},  // 42: 128 x 128, BK64, 3 stages, 4 + 4 waves
    {2, 2, 2, 2, 64, 4},   // 43: 128 x 128, BK32, 4 stages, 4 + 4 waves
    {2, 2, 2, 4, 64, 4},   // 44: 128 x 256, BK32, 4 stages, 4 + 4 waves
    {4, 1, 1, 2, 128, 3},  // 45: 128 x  64, BK64, 3 stages, 4 + 4 waves
    // high-occupancy 2-stage variants (conv_igemm.h ALIAS): epilogue scratch inside the idle ring stage
    {2, 2, 2, 2, 64, 2},   // 46: 128 x 128, BK32 (32 KB LDS, 4 workgroups per CU)
    {4, 1, 1, 2, 64, 2},   // 47: 128 x  64, BK32 (24 KB LDS)
    {4, 1, 1, 2, 128, 2},  // 48: 128 x  64, BK64 (48 KB LDS, 3 workgroups per CU)
    {2, 2, 1, 2, 128, 2},  // 49:  64 x 128, BK64 (48 KB LDS)
    // tile widths for the channel counts of yolov5m (multiples of 96) and yolov5x (multiples of 160): no padded filter rows
    {2, 2, 2, 5, 64, 2},   // 50: 128 x 320, BK32
    {4, 1, 1, 5, 128, 2},  // 51: 128 x 160, BK64
    {2, 2, 2, 3, 64, 2},   // 52: 128 x 192, BK32
    {4, 1, 1, 3, 128, 2},  // 53: 128 x  96, BK64
    {4, 2, 2, 5, 64, 2},   // 54: 256 x 320, BK32, 8 waves
    {4, 2, 2, 3, 64, 2},   // 55: 256 x 192, BK32, 8 waves
};
constexpr TileCfg kCfgs[kNumIgemm] = {
    {4, 1, 1, 1, 64},   //  0: 128 x  32, BK32
    {4, 1, 1, 2, 64},   //  1: 128 x  64, BK32
    {2, 2, 2, 2, 64},   //  2: 128 x 128, BK32
    {2, 2, 2, 4, 64},   //  3: 128 x 256, BK32
    {4, 1, 2, 1, 64},   //  4: 256 x  32, BK32
    {4, 1, 2, 2, 64},   //  5: 256 x  64, BK32
    {4, 1, 1, 1, 128},  //  6: 128 x  32, BK64
    {4, 1, 1, 2, 128},  //  7: 128 x  64, BK64
    {2, 2, 2, 2, 128},  //  8: 128 x 128, BK64
    {2, 2, 2, 4, 128},  //  9: 128 x 256, BK64
    {4, 1, 2, 2, 128},  // 10: 256 x  64, BK64
    {2, 2, 1, 2, 128},  // 11:  64 x 128, BK64
    {2, 2, 4, 2, 128},  // 12: 256 x 128, BK64
    {4, 1, 2, 1, 128},  // 13: 256 x  32, BK64
};


// stream-K configurations (conv_igemm.h SK): ids kSk0 + index; (wm, wn, tm, tn), 128-byte LDS rows (BK64), 2 stages
constexpr int kSk0 = 57, kNumSk = 4;
constexpr TileCfg kSkCfgs[kNumSk] = {
    {2, 2, 2, 2, 128},  // 57: 128 x 128
    {2, 4, 4, 2, 128},  // 58: 256 x 256, 8 waves
    {2, 2, 4, 2, 128},  // 59: 256 x 128
    {2, 2, 2, 4, 128},  // 60: 128 x 256
};
constexpr size_t kSkMaxGrid = 1024;                                   // workgroups a stream-K launch may have
constexpr size_t kSkSlabBytes = (size_t)256 * 256 * 4;                // largest tile, fp32
constexpr size_t kSkFlagBytes = kSkMaxGrid * 4;
struct SkWs { void* ws; size_t bytes; };
SkWs g_sk[64] = {};

template <typename T, int WM, int WN, int TM, int TN, int RB, bool TABLE, int NS = 2, bool PROD = false, bool ALIAS = false, bool SK = false, bool UP2 = false>
int launch_cfg(const Y5ConvParams& p0, int max_blocks, hipStream_t stream) {
  using Gm = Y5ConvGeom<T, RB>;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  Y5ConvParams p = p0;
  p.tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.Npad + BN - 1) / BN;
  p.nk = (p.K + Gm::BK - 1) / Gm::BK;
  y5_conv_set_fastdiv(p);
  if (p.nk * Gm::BK > p.Kpad) return y5_fail(Y5_ERR_BAD_ARG, "conv: Kpad smaller than K rounded up to the tile config's K per stage");
  int pieces = 0;
  if (TABLE) {
    pieces = p.Kpad / Gm::EPP;
    if (pieces > Y5_CONV_MAXTAB) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: K too large for gather-table mode");
  }
  const size_t lds = y5_conv_lds_bytes<T, WM, WN, TM, TN, RB, NS, ALIAS>(pieces);
  if (lds > 160 * 1024) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: tile configuration exceeds 160 KiB of LDS");
  auto kern = y5_conv_igemm_kernel<T, WM, WN, TM, TN, RB, TABLE, NS, PROD, ALIAS, SK, UP2>;
  constexpr int NTHREADS = WM * WN * 64 * (PROD ? 2 : 1);
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long ntiles = (long long)p.tilesM * p.tilesN;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return y5_fail(Y5_ERR_BAD_ARG, "conv: grid out of range");
  // persistent grid: as many workgroups as stay resident (CUs x occupancy), each walking ntiles/G tiles
  long long G = max_blocks;
  if (G <= 0) {
    const int g_num_cu = y5_num_cu();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), NTHREADS, lds) != hipSuccess || occ < 1)
      occ = 1;
    G = (long long)g_num_cu * occ;
  }
  if constexpr (SK) {
    // every resident workgroup gets an equal share of the tiles * nk chunk-units (at least two chunks each)
    const long long U = ntiles * p.nk;
    if (G > U / 2) G = U / 2 > 0 ? U / 2 : 1;
    if (G > (long long)kSkMaxGrid) G = kSkMaxGrid;
    int dev = 0;
    hipGetDevice(&dev);
    const SkWs& w = g_sk[dev < 64 ? dev : 0];
    const size_t need = kSkFlagBytes + (size_t)G * BM * BN * 4;
    if (!w.ws || w.bytes < need) return y5_fail(Y5_ERR_WORKSPACE, "conv: stream-K configuration needs y5_conv_set_sk_workspace()");
    p.sk_flags = static_cast<unsigned*>(w.ws);
    p.sk_ws = reinterpret_cast<float*>(static_cast<char*>(w.ws) + kSkFlagBytes);
    if (p.o_mul_h) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: stream-K with output placement is not built");
  } else {
    if (G > ntiles) G = ntiles;
    if (G >= 8) G &= ~7LL;  // y5_xcd_remap of the virtual block id needs G % 8 == 0 when blocks own several tiles
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(NTHREADS), lds, stream, p);
  return y5_check_launch("y5_conv2d_fwd");
}

template <typename T, bool TABLE>
int launch_by_cfg(const Y5ConvParams& p, int cfg, int mb, hipStream_t s) {
  if constexpr (sizeof(T) == 4) {
    switch (cfg) {
      case 0: return launch_cfg<T, 4, 1, 1, 1, 64, TABLE>(p, mb, s);
      case 1: return launch_cfg<T, 4, 1, 1, 2, 64, TABLE>(p, mb, s);
      case 2: return launch_cfg<T, 2, 2, 2, 2, 64, TABLE>(p, mb, s);
      case 3: return launch_cfg<T, 2, 2, 2, 4, 64, TABLE>(p, mb, s);
    }
    return y5_fail(Y5_ERR_BAD_ARG, "conv: fp32 supports tile configs 0..3 only");
  } else {
    switch (cfg) {
      case 0: return launch_cfg<T, 4, 1, 1, 1, 64, TABLE>(p, mb, s);
      case 1: return launch_cfg<T, 4, 1, 1, 2, 64, TABLE>(p, mb, s);
      case 2: return launch_cfg<T, 2, 2, 2, 2, 64, TABLE>(p, mb, s);
      case 3: return launch_cfg<T, 2, 2, 2, 4, 64, TABLE>(p, mb, s);
      case 4: return launch_cfg<T, 4, 1, 2, 1, 64, TABLE>(p, mb, s);
      case 5: return launch_cfg<T, 4, 1, 2, 2, 64, TABLE>(p, mb, s);
      case 6: return launch_cfg<T, 4, 1, 1, 1, 128, TABLE>(p, mb, s);
      case 7: return launch_cfg<T, 4, 1, 1, 2, 128, TABLE>(p, mb, s);
      case 8: return launch_cfg<T, 2, 2, 2, 2, 128, TABLE>(p, mb, s);
      case 9: return launch_cfg<T, 2, 2, 2, 4, 128, TABLE>(p, mb, s);
      case 10: return launch_cfg<T, 4, 1, 2, 2, 128, TABLE>(p, mb, s);
      case 11: return launch_cfg<T, 2, 2, 1, 2, 128, TABLE>(p, mb, s);
      case 12: return launch_cfg<T, 2, 2, 4, 2, 128, TABLE>(p, mb, s);
      case 13: return launch_cfg<T, 4, 1, 2, 1, 128, TABLE>(p, mb, s);
      // 3-stage LDS ring (counted vmcnt, one raw barrier per chunk): ids kRing0 + 0..7
      case kRing0 + 0: return launch_cfg<T, 4, 1, 1, 2, 64, TABLE, 3>(p, mb, s);
      case kRing0 + 1: return launch_cfg<T, 2, 2, 2, 2, 64, TABLE, 3>(p, mb, s);
      case kRing0 + 2: return launch_cfg<T, 2, 2, 2, 4, 64, TABLE, 3>(p, mb, s);
      case kRing0 + 3: return launch_cfg<T, 4, 1, 2, 2, 64, TABLE, 3>(p, mb, s);
      case kRing0 + 4: return launch_cfg<T, 4, 1, 1, 2, 128, TABLE, 3>(p, mb, s);
      case kRing0 + 5: return launch_cfg<T, 2, 2, 2, 2, 128, TABLE, 3>(p, mb, s);
      case kRing0 + 6: return launch_cfg<T, 2, 2, 1, 2, 128, TABLE, 3>(p, mb, s);
      case kRing0 + 7: return launch_cfg<T, 2, 2, 4, 2, 128, TABLE, 3>(p, mb, s);
      case kBig0 + 0: return launch_cfg<T, 2, 2, 4, 2, 64, TABLE, 4>(p, mb, s);
      case kBig0 + 1: return launch_cfg<T, 4, 2, 2, 2, 64, TABLE, 4>(p, mb, s);
      case kBig0 + 2: return launch_cfg<T, 2, 4, 4, 2, 64, TABLE, 4>(p, mb, s);
      case kBig0 + 3: return launch_cfg<T, 2, 4, 4, 2, 64, TABLE, 3>(p, mb, s);
      case kBig0 + 4: return launch_cfg<T, 2, 4, 4, 2, 128, TABLE, 2>(p, mb, s);
      case kBig0 + 5: return launch_cfg<T, 2, 2, 4, 2, 128, TABLE, 3, true>(p, mb, s);
      case kBig0 + 6: return launch_cfg<T, 2, 2, 4, 2, 64, TABLE, 4, true>(p, mb, s);
      case kBig0 + 7: return launch_cfg<T, 2, 2, 2, 2, 128, TABLE, 3, true>(p, mb, s);
      case kBig0 + 8: return launch_cfg<T, 2, 2, 2, 2, 64, TABLE, 4, true>(p, mb, s);
      case kBig0 + 9: return launch_cfg<T, 2, 2, 2, 4, 64, TABLE, 4, true>(p, mb, s);
      case kBig0 + 10: return launch_cfg<T, 4, 1, 1, 2, 128, TABLE, 3, true>(p, mb, s);
      case kBig0 + 11: return launch_cfg<T, 2, 2, 2, 2, 64, TABLE, 2, false, true>(p, mb, s);
      case kBig0 + 12: return launch_cfg<T, 4, 1, 1, 2, 64, TABLE, 2, false, true>(p, mb, s);
      case kBig0 + 13: return launch_cfg<T, 4, 1, 1, 2, 128, TABLE, 2, false, true>(p, mb, s);
      case kBig0 + 14: return launch_cfg<T, 2, 2, 1, 2, 128, TABLE, 2, false, true>(p, mb, s);
      case kBig0 + 15: return launch_cfg<T, 2, 2, 2, 5, 64, TABLE, 2>(p, mb, s);
      case kBig0 + 16: return launch_cfg<T, 4, 1, 1, 5, 128, TABLE, 2>(p, mb, s);
      case kBig0 + 17: return launch_cfg<T, 2, 2, 2, 3, 64, TABLE, 2>(p, mb, s);
      case kBig0 + 18: return launch_cfg<T, 4, 1, 1, 3, 128, TABLE, 2>(p, mb, s);
      case kBig0 + 19: return launch_cfg<T, 4, 2, 2, 5, 64, TABLE, 2>(p, mb, s);
      case kBig0 + 20: return launch_cfg<T, 4, 2, 2, 3, 64, TABLE, 2>(p, mb, s);
      case kSk0 + 0: return launch_cfg<T, 2, 2, 2, 2, 128, TABLE, 2, false, false, true>(p, mb, s);
      case kSk0 + 1: return launch_cfg<T, 2, 4, 4, 2, 128, TABLE, 2, false, false, true>(p, mb, s);
      case kSk0 + 2: return launch_cfg<T, 2, 2, 4, 2, 128, TABLE, 2, false, false, true>(p, mb, s);
      case kSk0 + 3: return launch_cfg<T, 2, 2, 2, 4, 128, TABLE, 2, false, false, true>(p, mb, s);
    }
    return y5_fail(Y5_ERR_BAD_ARG, "conv: unknown tile config");
  }
}

// ---- streaming pointwise configurations (conv_pw.h): id = kNumIgemm + index ---------------------------------
struct PwCfg { int kc, rb, nt, s; };
constexpr int kNumPw = 13;
constexpr int kPw8_0 = 84;  // ids 84.. = kPwCfgs[9..]: eight waves per workgroup, one stage per wave
// ids 88, 89: implicit GEMM whose loader reads `nn.Upsample(2) + Concat` virtually (conv_igemm.h UP2; 1x1 s1 layers with d->up_c > 0 ONLY):
// 88 = the producer / consumer 128 x 128 BK32 ring of id 43, 89 = the plain 2-stage 128 x 128 BK64 tile of id 8
constexpr int kUp0 = 88, kNumUp = 2;
constexpr int kPwk0 = 93;                  // ids 93, 94 = K-streamed pointwise kernel of conv_pwk.h (256- / 128-channel N tile)
constexpr int kG8_0 = 95;                  // ids 95.. = 256-row / 8-phase implicit GEMM of conv_g8.h (convg8.hip)
constexpr int kH3S_0 = 90, kNumH3a = 17;   // ids 90.. = halo-resident 3x3 configurations 17.. of convh3.hip (small tiles for the stride-2 layers)
constexpr TileCfg kUpCfgs[kNumUp] = {{2, 2, 2, 2, 64}, {2, 2, 2, 2, 128}};
constexpr int kPw2_0 = 56;  // pointwise configurations added after the id space was laid out: ids 56.. = kPwCfgs[8..]
constexpr PwCfg kPwCfgs[kNumPw] = {
    {1, 64, 1, 4},   // 14:  32 ->  32, 4 stages
    {1, 128, 1, 4},  // 15:  64 ->  32
    {1, 128, 2, 4},  // 16:  64 ->  64
    {1, 128, 2, 3},  // 17:  64 ->  64, 3 stages
    {2, 128, 2, 3},  // 18: 128 ->  64
    {2, 128, 4, 3},  // 19: 128 -> 128
    {2, 128, 4, 2},  // 20: 128 -> 128, 2 stages
    {2, 128, 2, 4},  // 21: 128 ->  64, 4 stages
    {2, 128, 8, 2},  // 56: 128 -> 256 (the P3 Detect head), epilogue in two channel groups
    {2, 128, 4, 1},  // 84: 128 -> 128, eight waves
    {1, 128, 2, 1},  // 85:  64 ->  64, eight waves
    {2, 128, 2, 1},  // 86: 128 ->  64, eight waves
    {2, 128, 8, 1},  // 87: 128 -> 256, eight waves, epilogue in two channel groups
};

template <int KC, int RB, int NT, int S, bool UP2, bool ACT, int OS = 1, int NWV = 4>
int launch_pw_v(const Y5ConvParams& p, int max_blocks, hipStream_t stream) {
  const size_t lds = y5_conv_pw_lds_bytes<KC, RB, NT, S, OS, NWV>();
  if (lds > 160 * 1024) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: pointwise configuration exceeds 160 KiB of LDS");
  auto kern = y5_conv_pw_kernel<KC, RB, NT, S, UP2, ACT, OS, NWV>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long nbt = ((long long)(p.M >> 5) + NWV - 1) / NWV;
  long long G = max_blocks;
  if (G <= 0) {
    const int g_num_cu = y5_num_cu();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), NWV * 64, lds) != hipSuccess || occ < 1) occ = 1;
    G = (long long)g_num_cu * occ;
  }
  if (G > nbt) G = nbt;
  if (G >= 8) G &= ~7LL;
  if (p.bn_partial) {
    if ((size_t)G * 2 * p.C2 * 4 > p.bn_bytes) return y5_fail(Y5_ERR_WORKSPACE, "conv_fwd_stats: partial buffer too small for this grid");
    *p.bn_rows = (int)G;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(NWV * 64), lds, stream, p);
  return y5_check_launch("y5_conv2d_fwd(pw)");
}
template <int KC, int RB, int NT, int S, int OS = 1, int NWV = 4>
int launch_pw(const Y5ConvParams& p, int mb, hipStream_t st) {
  if constexpr (OS == 1) {
    if (p.y2 && !p.split_n) return p.act ? launch_pw_v<KC, RB, NT, S, true, true, 1, NWV>(p, mb, st) : launch_pw_v<KC, RB, NT, S, true, false, 1, NWV>(p, mb, st);
  } else {
    if (p.y2 && !p.split_n) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: split-epilogue pointwise configuration has no upsampled replica");
  }
  return p.act ? launch_pw_v<KC, RB, NT, S, false, true, OS, NWV>(p, mb, st) : launch_pw_v<KC, RB, NT, S, false, false, OS, NWV>(p, mb, st);
}

int launch_pw_by_cfg(const Y5ConvParams& p, int idx, int mb, hipStream_t s) {
  switch (idx) {
    case 0: return launch_pw<1, 64, 1, 4>(p, mb, s);
    case 1: return launch_pw<1, 128, 1, 4>(p, mb, s);
    case 2: return launch_pw<1, 128, 2, 4>(p, mb, s);
    case 3: return launch_pw<1, 128, 2, 3>(p, mb, s);
    case 4: return launch_pw<2, 128, 2, 3>(p, mb, s);
    case 5: return launch_pw<2, 128, 4, 3>(p, mb, s);
    case 6: return launch_pw<2, 128, 4, 2>(p, mb, s);
    case 7: return launch_pw<2, 128, 2, 4>(p, mb, s);
    case 8: return launch_pw<2, 128, 8, 2, 2>(p, mb, s);
    case 9: return launch_pw<2, 128, 4, 1, 1, 8>(p, mb, s);
    case 10: return launch_pw<1, 128, 2, 1, 1, 8>(p, mb, s);
    case 11: return launch_pw<2, 128, 2, 1, 1, 8>(p, mb, s);
    case 12: return launch_pw<2, 128, 8, 1, 2, 8>(p, mb, s);
  }
  return y5_fail(Y5_ERR_BAD_ARG, "conv: unknown pointwise config");
}

// ---- streaming 3x3 configurations (conv_k3.h): ids kK3_0 + index ------------------------------------------------
struct K3Cfg { int c1, nt, sh, s; };
constexpr int kK3_0 = 30, kNumK3 = 11;
constexpr int kK3W_0 = 78;  // ids 78, 79 = kK3Cfgs[5], [6]: the 64-channel kernel with the filter in registers (added after the id space was laid out)
constexpr K3Cfg kK3Cfgs[kNumK3] = {
    {32, 1, 1, 3},  // 30: 3x3 s1 32->32, 3 stages   (Bottleneck.cv2 @160)
    {32, 2, 2, 2},  // 31: 3x3 s2 32->64, 2 stages   (Conv 1 @320->160)
    {64, 2, 1, 2},  // 32: 3x3 s1 64->64, 2 stages   (Bottleneck.cv2 @80)
    {32, 1, 1, 2},  // 33: 3x3 s1 32->32, 2 stages
    {32, 2, 2, 3},  // 34: 3x3 s2 32->64, 3 stages
    {64, 2, 1, 3},  // 78: 3x3 s1 64->64, filter fragments in registers, 3 stages
    {64, 2, 1, 4},  // 79: the same, 4 stages
    {64, 2, 1, 1},  // 80: 3x3 s1 64->64, EIGHT waves with one stage each (two waves per SIMD under one LDS filter copy)
    {32, 2, 2, 1},  // 81: 3x3 s2 32->64, eight waves, one stage
    {32, 1, 1, 1},  // 82: 3x3 s1 32->32, eight waves, one stage
    {32, 1, 1, 2},  // 83: 3x3 s1 32->32, eight waves, two stages
};

template <int C1, int NT, int SH, int S, bool RES, bool ACT, int NT2 = 0, bool WREG = false, int NWV = 4>
int launch_k3_v(const Y5ConvParams& p, int max_blocks, hipStream_t stream) {
  const size_t lds = y5_conv_k3_lds_bytes<C1, NT, SH, S, NT2, WREG, NWV>();
  if (lds > 160 * 1024) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: 3x3 streaming configuration exceeds 160 KiB of LDS");
  auto kern = y5_conv_k3_kernel<C1, NT, SH, S, RES, ACT, NT2, WREG, NWV>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long nwt = (long long)p.B * (p.OH / 4) * (p.OW / 8);
  const long long nbt = (nwt + NWV - 1) / NWV;
  long long G = max_blocks;
  if (G <= 0) {
    const int g_num_cu = y5_num_cu();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), NWV * 64, lds) != hipSuccess || occ < 1) occ = 1;
    G = (long long)g_num_cu * occ;
  }
  if (G > nbt) G = nbt;
  if (G >= 8) G &= ~7LL;
  if (p.bn_partial) {
    if ((size_t)G * 2 * p.C2 * 4 > p.bn_bytes) return y5_fail(Y5_ERR_WORKSPACE, "conv_fwd_stats: partial buffer too small for this grid");
    *p.bn_rows = (int)G;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(NWV * 64), lds, stream, p);
  return y5_check_launch("y5_conv2d_fwd(k3)");
}
template <int C1, int NT, int SH, int S, bool WREG = false, int NWV = 4>
int launch_k3(const Y5ConvParams& p, int mb, hipStream_t st) {
  if (p.res) return p.act ? launch_k3_v<C1, NT, SH, S, true, true, 0, WREG, NWV>(p, mb, st) : launch_k3_v<C1, NT, SH, S, true, false, 0, WREG, NWV>(p, mb, st);
  return p.act ? launch_k3_v<C1, NT, SH, S, false, true, 0, WREG, NWV>(p, mb, st) : launch_k3_v<C1, NT, SH, S, false, false, 0, WREG, NWV>(p, mb, st);
}

int launch_k3_by_cfg(const Y5ConvParams& p, int idx, int mb, hipStream_t s) {
  switch (idx) {
    case 5: return launch_k3<64, 2, 1, 3, true>(p, mb, s);
    case 6: return launch_k3<64, 2, 1, 4, true>(p, mb, s);
    case 7: return launch_k3<64, 2, 1, 1, false, 8>(p, mb, s);
    case 8: return launch_k3<32, 2, 2, 1, false, 8>(p, mb, s);
    case 9: return launch_k3<32, 1, 1, 1, false, 8>(p, mb, s);
    case 10: return launch_k3<32, 1, 1, 2, false, 8>(p, mb, s);
    case 0: return launch_k3<32, 1, 1, 3>(p, mb, s);
    case 1: return launch_k3<32, 2, 2, 2>(p, mb, s);
    case 2: return launch_k3<64, 2, 1, 2>(p, mb, s);
    case 3: return launch_k3<32, 1, 1, 2>(p, mb, s);
    case 4: return launch_k3<32, 2, 2, 3>(p, mb, s);
  }
  return y5_fail(Y5_ERR_BAD_ARG, "conv: unknown 3x3 streaming config");
}

}  // namespace
int y5_launch_h3_by_cfg(const Y5ConvParams& p, int idx, int mb, hipStream_t s);
int y5_launch_pwk_by_cfg(const Y5ConvParams& p, int idx, hipStream_t s);
int y5_launch_g8_by_cfg(const Y5ConvParams& p, int idx, int mb, hipStream_t s);
void y5_h3_cfg_info(int idx, int* bm, int* bn);
namespace {
constexpr int kH3_0 = 61;

int default_cfg(const y5_conv_desc* d) {
  const int n = d->Npad;
  if (d->dtype == Y5_F32) return n <= 32 ? 0 : n <= 64 ? 1 : 2;
  const bool k64 = d->Kpad % 64 == 0 && d->C1 % 64 == 0;
  if (n <= 32) return k64 ? 6 : 0;
  if (n <= 64) return k64 ? 7 : 1;
  return k64 ? 8 : 2;
}

}  // namespace

extern "C" int y5_conv_num_cfgs(void) { return Y5_CONV_NUM_CFGS; }

// flags + the largest slab set: 1024 workgroups x 128x128 = 512 x 256x128 = 256 x 256x256 fp32 tiles = 64 MiB
extern "C" size_t y5_conv_sk_workspace_bytes(void) { return kSkFlagBytes + kSkMaxGrid * (size_t)128 * 128 * 4; }

extern "C" int y5_conv_set_sk_workspace(void* ws, size_t bytes, void* stream_) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return y5_fail(Y5_ERR_RUNTIME, "conv_set_sk_workspace: no device");
  if (!ws) { g_sk[dev] = SkWs{nullptr, 0}; return Y5_OK; }
  if (((uintptr_t)ws & 255) || bytes < kSkFlagBytes + kSkSlabBytes) return y5_fail(Y5_ERR_BAD_ARG, "conv_set_sk_workspace: misaligned or too small");
  if (hipMemsetAsync(ws, 0, kSkFlagBytes, static_cast<hipStream_t>(stream_)) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "conv_set_sk_workspace: memset failed");
  g_sk[dev] = SkWs{ws, bytes};
  return Y5_OK;
}

extern "C" int y5_conv_cfg_info(int cfg, int* bm, int* bn, int* bk_bytes) {
  if (cfg < 0 || cfg >= Y5_CONV_NUM_CFGS) return y5_fail(Y5_ERR_BAD_ARG, "conv_cfg_info: bad id");
  if (cfg >= kG8_0) {
    if (bm) *bm = 256;
    if (bn) *bn = cfg == kG8_0 ? 256 : 128;
    if (bk_bytes) *bk_bytes = 128;
    return Y5_OK;
  }
  if (cfg >= kPwk0) {
    if (bm) *bm = 256;
    if (bn) *bn = cfg == kPwk0 ? 256 : 128;
    if (bk_bytes) *bk_bytes = 64;
    return Y5_OK;
  }
  if (cfg >= kH3S_0) {
    int m = 0, n = 0;
    y5_h3_cfg_info(kNumH3a + cfg - kH3S_0, &m, &n);
    if (bm) *bm = m;
    if (bn) *bn = n;
    if (bk_bytes) *bk_bytes = 64;
    return Y5_OK;
  }
  if (cfg >= kUp0) {
    const TileCfg& c = kUpCfgs[cfg - kUp0];
    if (bm) *bm = c.wm * c.tm * 32;
    if (bn) *bn = c.wn * c.tn * 32;
    if (bk_bytes) *bk_bytes = c.rb;
    return Y5_OK;
  }
  if (cfg >= kPw8_0) {
    const PwCfg& c = kPwCfgs[9 + cfg - kPw8_0];
    if (bm) *bm = 256;
    if (bn) *bn = c.nt * 32;
    if (bk_bytes) *bk_bytes = c.kc * c.rb;
    return Y5_OK;
  }
  if (cfg >= kK3W_0) {
    const K3Cfg& c = kK3Cfgs[5 + cfg - kK3W_0];
    if (bm) *bm = 128;
    if (bn) *bn = c.nt * 32;
    if (bk_bytes) *bk_bytes = 9 * c.c1 * 2;
    return Y5_OK;
  }
  if (cfg >= kH3_0) {
    int m = 0, n = 0;
    y5_h3_cfg_info(cfg - kH3_0, &m, &n);
    if (bm) *bm = m;
    if (bn) *bn = n;
    if (bk_bytes) *bk_bytes = 64;
    return Y5_OK;
  }
  if (cfg >= kSk0) {
    const TileCfg& c = kSkCfgs[cfg - kSk0];
    if (bm) *bm = c.wm * c.tm * 32;
    if (bn) *bn = c.wn * c.tn * 32;
    if (bk_bytes) *bk_bytes = c.rb;
    return Y5_OK;
  }
  if (cfg >= kPw2_0) {
    const PwCfg& c = kPwCfgs[8 + cfg - kPw2_0];
    if (bm) *bm = 128;
    if (bn) *bn = c.nt * 32;
    if (bk_bytes) *bk_bytes = c.kc * c.rb;
    return Y5_OK;
  }
  if (cfg >= kBig0) {
    const BigCfg& c = kBigCfgs[cfg - kBig0];
    if (bm) *bm = c.wm * c.tm * 32;
    if (bn) *bn = c.wn * c.tn * 32;
    if (bk_bytes) *bk_bytes = c.rb;
    return Y5_OK;
  }
  if (cfg >= kK3_0) {
    const K3Cfg& c = kK3Cfgs[cfg - kK3_0];
    if (bm) *bm = 128;
    if (bn) *bn = c.nt * 32;
    if (bk_bytes) *bk_bytes = 9 * c.c1 * 2;
    return Y5_OK;
  }
  if (cfg >= kRing0) cfg = kRingBase[cfg - kRing0];
  if (cfg >= kNumIgemm) {
    const PwCfg& c = kPwCfgs[cfg - kNumIgemm];
    if (bm) *bm = 128;
    if (bn) *bn = c.nt * 32;
    if (bk_bytes) *bk_bytes = c.kc * c.rb;
    return Y5_OK;
  }
  if (bm) *bm = kCfgs[cfg].wm * kCfgs[cfg].tm * 32;
  if (bn) *bn = kCfgs[cfg].wn * kCfgs[cfg].tn * 32;
  if (bk_bytes) *bk_bytes = kCfgs[cfg].rb;
  return Y5_OK;
}

static int conv2d_fwd_impl(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, void* y_up2,
                           float* stats_partial, size_t stats_bytes, int* stats_rows, void* stream_);

extern "C" int y5_conv2d_fwd(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                             const void* residual, void* y, void* y_up2, void* stream_) {
  return conv2d_fwd_impl(d, x, w_packed, bias, residual, y, y_up2, nullptr, 0, nullptr, stream_);
}

// Train-mode forward of a Conv block's convolution (models/common.py:82-88: act(bn(conv(x))), conv without bias / activation) that ALSO leaves the
// per-channel batch statistics of its output as per-workgroup partials -- see include/yolov5_hip.h
extern "C" int y5_conv2d_fwd_stats(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y, float* partial,
                                   size_t partial_bytes, int* rows, void* stream_) {
  if (!partial || !rows || ((uintptr_t)partial & 15)) return y5_fail(Y5_ERR_BAD_ARG, "conv_fwd_stats: partial / rows must be given (partial 16-byte aligned)");
  *rows = 0;
  return conv2d_fwd_impl(d, x, w_packed, bias, nullptr, y, nullptr, partial, partial_bytes, rows, stream_);
}

static int conv2d_fwd_impl(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, void* y_up2,
                           float* stats_partial, size_t stats_bytes, int* stats_rows, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!d || !x || !w_packed || !bias || (!y && !y_up2)) return y5_fail(Y5_ERR_BAD_ARG, "conv: null pointer");
  const int es = d->dtype == Y5_F16 ? 2 : d->dtype == Y5_F32 ? 4 : 0;
  if (!es) return y5_fail(Y5_ERR_BAD_ARG, "conv: dtype must be Y5_F16 or Y5_F32");
  const int epp = 16 / es;
  int cfg = d->cfg < 0 ? (d->up_c > 0 ? kUp0 + 1 : default_cfg(d)) : d->cfg;
  if (cfg >= Y5_CONV_NUM_CFGS) return y5_fail(Y5_ERR_BAD_ARG, "conv: unknown tile config");
  const bool g8 = cfg >= kG8_0;           // 256-row / 8-phase implicit GEMM (ids 95..)
  const bool pwk = cfg >= kPwk0 && !g8;   // K-streamed pointwise kernel (ids 93, 94)
  const bool h3s = cfg >= kH3S_0 && !pwk && !g8; // halo-resident 3x3 configurations added in round 5 (ids 90..92)
  const bool up2 = cfg >= kUp0 && !h3s && !pwk && !g8;   // virtual upsample + concat loader (ids 88, 89)
  const bool up8 = g8 && d->up_c > 0;     // ... which the 8-phase family's loader reads as well (round 6)
  if (up2 != (d->up_c > 0) && !up8) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: only configurations 88 / 89 / 95 / 96 serve the layers with up_c > 0 (virtual upsample + concat), 88 / 89 no others");
  if (up2 || up8) {
    const int bkb = up8 ? 64 : kUpCfgs[cfg - kUp0].rb / 2;
    if (d->dtype != Y5_F16 || d->KH != 1 || d->KW != 1 || d->SH != 1 || d->SW != 1 || d->PH || d->PW || !residual || !y || (y_up2 != nullptr) != (d->split_n > 0) ||
        d->out_mul_h || (d->H & 1) || (d->W & 1) || d->up_c % bkb || d->up_c >= d->C1 || d->C1 % bkb || d->ld_up < d->up_c || (d->ld_up & 7))
      return y5_fail(Y5_ERR_UNSUPPORTED, "conv: virtual upsample + concat needs a 1x1 s1 fp16 layer on an even H x W grid, up_c and C1 multiples of the K chunk, "
                                         "the low-resolution tensor in the `residual` argument");
  }
  const int fam = up2 || pwk || g8 ? 2 : h3s ? kH3_0 : cfg;   // (88 / 89 are implicit-GEMM tiles; 90.. belong to the halo family: none of the id-range tests below applies to them)
  const bool pw8 = fam >= kPw8_0;         // streaming pointwise with eight waves per workgroup (ids 84..)
  const bool k3w = fam >= kK3W_0 && !pw8;         // streaming 3x3 added after the id space was laid out (ids 78..83)
  const bool h3 = fam >= kH3_0 && !k3w && !pw8;
  const bool sk = fam >= kSk0 && !h3 && !k3w && !pw8;
  const bool pw = (fam >= kNumIgemm && fam < kRing0) || (fam >= kPw2_0 && !sk && !h3 && !k3w);
  const int pwi = pw8 ? 9 + fam - kPw8_0 : fam >= kPw2_0 ? 8 + fam - kPw2_0 : fam - kNumIgemm;
  const bool big = fam >= kBig0 && fam < kPw2_0;
  const bool k3 = (fam >= kK3_0 && fam < kBig0) || k3w;
  const int k3i = k3w ? 5 + fam - kK3W_0 : fam - kK3_0;
  const int bk = up2 ? kUpCfgs[cfg - kUp0].rb / es : (pw || k3 || h3) ? 8 : (sk ? kSkCfgs[fam - kSk0].rb : big ? kBigCfgs[fam - kBig0].rb : kCfgs[fam >= kRing0 ? kRingBase[fam - kRing0] : fam].rb) / es;
  if (fam >= kRing0 && d->dtype != Y5_F16) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: ring configurations are fp16 only");
  if (d->C1 % epp || d->ldx % epp) return y5_fail(Y5_ERR_BAD_ARG, "conv: C1 and ldx must be multiples of 16 bytes");
  if (d->C2 % epp || (y && d->ldy % epp) || (residual && d->ldr % epp) || (y_up2 && d->ld2 % epp))
    return y5_fail(Y5_ERR_BAD_ARG, "conv: C2/ldy/ldr/ld2 must be multiples of 16 bytes");
  if (d->Kpad % epp || d->Kpad < d->KH * d->KW * d->C1 || d->Npad % 32 || d->Npad < d->C2)
    return y5_fail(Y5_ERR_BAD_ARG, "conv: bad packed filter dims (Kpad % 16 bytes, Kpad >= K, Npad % 32)");
  if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)y_up2) & 15)
    return y5_fail(Y5_ERR_BAD_ARG, "conv: pointers must be 16-byte aligned");
  const bool placed = d->out_mul_h != 0;
  if (placed && (d->out_mul_w < 1 || d->out_mul_h < 1 || d->out_off_h < 0 || d->out_off_w < 0 || d->OH < 1 || d->OW < 1 ||
                 (d->OH - 1) * d->out_mul_h + d->out_off_h >= d->out_H || (d->OW - 1) * d->out_mul_w + d->out_off_w >= d->out_W || y_up2))
    return y5_fail(Y5_ERR_BAD_ARG, "conv: bad output placement");
  const int oh = placed ? d->OH : (d->H + 2 * d->PH - d->KH) / d->SH + 1, ow = placed ? d->OW : (d->W + 2 * d->PW - d->KW) / d->SW + 1;
  if (oh != d->OH || ow != d->OW) return y5_fail(Y5_ERR_BAD_ARG, "conv: OH/OW inconsistent with H/W/k/s/p");
  if ((long long)d->B * d->H * d->W * d->ldx * es >= 0x7fffffffLL || (long long)d->B * oh * ow >= 0x7fffffffLL ||
      (long long)d->Npad * d->Kpad * es >= 0x7fffffffLL)
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv: tensor exceeds 2^31 bytes (32-bit buffer offsets)");

  Y5ConvParams p{};
  p.x = x; p.w = w_packed; p.bias = bias; p.res = (up2 || up8) ? nullptr : residual; p.y = y; p.y2 = y_up2;
  if (up2 || up8) {   // (the low-resolution source travels in the `residual` argument: a layer of this kind has no residual)
    p.x2 = residual; p.ldx2 = d->ld_up; p.up_c = d->up_c;
    const long long b2 = (((long long)d->B * (d->H / 2) * (d->W / 2) - 1) * d->ld_up + d->up_c) * 2;
    if (b2 >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: low-resolution tensor exceeds 2^31 bytes");
    p.x2_bytes = (unsigned)b2;
  }
  p.zero = y5_zero_page();
  if (!p.zero) return y5_fail(Y5_ERR_RUNTIME, "conv: zero page allocation failed");
  p.B = d->B; p.H = d->H; p.W = d->W; p.C1 = d->C1; p.ldx = d->ldx;
  p.OH = oh; p.OW = ow; p.C2 = d->C2; p.ldy = d->ldy;
  p.KH = d->KH; p.KW = d->KW; p.SH = d->SH; p.SW = d->SW; p.PH = d->PH; p.PW = d->PW;
  p.act = d->act; p.Kpad = d->Kpad; p.Npad = d->Npad; p.K = d->KH * d->KW * d->C1;
  p.ldr = d->ldr; p.ld2 = d->ld2;
  p.split_n = d->split_n;
  if (d->split_n) {
    if (d->split_n < 0 || d->split_n % epp || d->split_n >= d->C2 || !y || !y_up2 || d->ld2 % epp || d->ld2 < d->C2 - d->split_n || d->ldy < d->split_n ||
        placed || (residual && !up2 && !up8))
      return y5_fail(Y5_ERR_BAD_ARG, "conv: bad split store (split_n multiple of 16 bytes inside C2, both destinations, no residual / placement)");
    if (k3 || h3) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: split store needs a pointwise / implicit-GEMM configuration");
  }
  p.M = d->B * oh * ow;
  p.o_mul_h = d->out_mul_h; p.o_mul_w = d->out_mul_w; p.o_off_h = d->out_off_h; p.o_off_w = d->out_off_w; p.o_H = d->out_H; p.o_W = d->out_W;
  if (placed && (pw || k3 || h3)) return y5_fail(Y5_ERR_UNSUPPORTED, "conv: output placement needs a general implicit-GEMM configuration");
  p.x_bytes = (unsigned)((((long long)d->B * d->H * d->W - 1) * d->ldx + d->C1) * es);
  p.w_bytes = (unsigned)((long long)d->Npad * d->Kpad * es);
  if (stats_partial) {   // fused BatchNorm statistics: the act = 0 instantiations of the streaming kernels carry them
    if (!(pw || k3) || d->act || d->dtype != Y5_F16 || d->split_n || placed || !y)
      return y5_fail(Y5_ERR_UNSUPPORTED, "conv_fwd_stats: needs a streaming pointwise / 3x3 configuration, fp16, act = 0, one destination");
    p.bn_partial = stats_partial; p.bn_bytes = stats_bytes; p.bn_rows = stats_rows;
  }

  if (g8) {
    if (d->dtype != Y5_F16 || !y || (y_up2 && !d->split_n) || placed || (d->C1 & 7) || d->C1 < 64 || (d->Kpad & 63) || d->KH * d->KW > 32 || d->Npad > 2048)
      return y5_fail(Y5_ERR_UNSUPPORTED, "conv: the 8-phase configurations need an fp16 layer with C1 % 8 == 0, C1 >= 64, Kpad % 64 == 0, at most 32 taps, Npad <= 2048, no replica / placement");
    return y5_launch_g8_by_cfg(p, cfg - kG8_0, d->max_blocks, stream);
  }
  if (pwk) {
    if (d->dtype != Y5_F16 || d->KH != 1 || d->KW != 1 || d->SH != 1 || d->SW != 1 || d->PH || d->PW || residual || !y || (y_up2 && !d->split_n) || placed ||
        (d->C1 & 31) || d->Kpad < d->C1)
      return y5_fail(Y5_ERR_UNSUPPORTED, "conv: the K-streamed pointwise configurations need a 1x1 s1 fp16 layer with C1 % 32 == 0, no residual / replica / placement");
    return y5_launch_pwk_by_cfg(p, cfg - kPwk0, stream);
  }
  if (h3) {
    if (d->dtype != Y5_F16 || d->KH != 3 || d->KW != 3 || d->SH != d->SW || (d->SH != 1 && d->SH != 2) || d->PH != 1 || d->PW != 1 || !y || y_up2 || d->C1 % 32 ||
        d->Kpad < 9 * d->C1 || placed)
      return y5_fail(Y5_ERR_UNSUPPORTED, "conv: halo 3x3 configuration needs a 3x3 s1 / s2 p1 fp16 layer with C1 % 32 == 0 and a single destination");
    return y5_launch_h3_by_cfg(p, h3s ? kNumH3a + cfg - kH3S_0 : cfg - kH3_0, d->max_blocks, stream);
  }
  if (k3) {
    const K3Cfg& c = kK3Cfgs[k3i];
    if (d->dtype != Y5_F16 || d->KH != 3 || d->KW != 3 || d->SH != c.sh || d->SW != c.sh || d->PH != 1 || d->PW != 1 || !y || y_up2 ||
        d->C1 != c.c1 || d->Npad != c.nt * 32 || (oh & 3) || (ow & 7) || d->Kpad < 9 * c.c1 || d->H > 255 * 4 || d->W > 65535)
      return y5_fail(Y5_ERR_UNSUPPORTED, "conv: 3x3 streaming configuration does not match this layer");
    return launch_k3_by_cfg(p, k3i, d->max_blocks, stream);
  }
  if (pw) {
    const PwCfg& c = kPwCfgs[pwi];
    if (d->dtype != Y5_F16 || d->KH != 1 || d->KW != 1 || d->SH != 1 || d->SW != 1 || d->PH || d->PW || residual || !y ||
        d->C1 != c.kc * c.rb / 2 || d->Npad != c.nt * 32 || (p.M & 31) || d->Kpad * 2 < c.kc * c.rb)
      return y5_fail(Y5_ERR_UNSUPPORTED, "conv: pointwise configuration does not match this layer");
    return launch_pw_by_cfg(p, pwi, d->max_blocks, stream);
  }
  if (up2) {
    if (cfg == kUp0) return launch_cfg<half_t, 2, 2, 2, 2, 64, false, 4, true, false, false, true>(p, d->max_blocks, stream);
    return launch_cfg<half_t, 2, 2, 2, 2, 128, false, 2, false, false, false, true>(p, d->max_blocks, stream);
  }
  const bool table = (d->C1 % bk) != 0 || d->KH * d->KW > 32;  // uniform mode keeps a 32-bit tap-validity mask per row
  if (d->dtype == Y5_F16)
    return table ? launch_by_cfg<half_t, true>(p, cfg, d->max_blocks, stream) : launch_by_cfg<half_t, false>(p, cfg, d->max_blocks, stream);
  return table ? launch_by_cfg<float, true>(p, cfg, d->max_blocks, stream) : launch_by_cfg<float, false>(p, cfg, d->max_blocks, stream);
}

extern "C" int y5_conv2d_time(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, const void* residual,
                              void* y, void* y_up2, int iters, void* stream_, float* ms) {
  if (!ms || iters < 1) return y5_fail(Y5_ERR_BAD_ARG, "conv_time: bad args");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  int rc = y5_conv2d_fwd(d, x, w_packed, bias, residual, y, y_up2, stream_);  // warm-up + validation
  if (rc) return rc;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "event create failed");
  hipEventRecord(e0, st);
  for (int i = 0; i < iters && !rc; ++i) rc = y5_conv2d_fwd(d, x, w_packed, bias, residual, y, y_up2, stream_);
  hipEventRecord(e1, st);
  if (hipEventSynchronize(e1) != hipSuccess) rc = y5_fail(Y5_ERR_RUNTIME, "event sync failed");
  else { hipEventElapsedTime(ms, e0, e1); *ms /= (float)iters; }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}

// ---- 3x3 s2 Conv + the pointwise convolution behind it as one launch (conv_k3.h, PW2) --------------------------------------------
extern "C" int y5_conv_k3pw_fwd(const y5_conv_desc* d, const void* x, const void* w1_packed, const float* bias1, const void* w2_packed,
                                const float* bias2, int C3, int Npad2, int Kpad2, int act2, void* y, int ldy, void* y2, int ld2, int split_n,
                                void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!d || !x || !w1_packed || !bias1 || !w2_packed || !bias2 || !y) return y5_fail(Y5_ERR_BAD_ARG, "conv_k3pw: null pointer");
  const int cfg = d->cfg < 0 ? kK3_0 + 4 : d->cfg;
  if (cfg != kK3_0 + 1 && cfg != kK3_0 + 4 && cfg != 81) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_k3pw: built for the 3x3 s2 32->64 streaming configurations (31, 34, 81)");
  const K3Cfg& c = kK3Cfgs[1];
  const int oh = (d->H + 2 - 3) / 2 + 1, ow = (d->W + 2 - 3) / 2 + 1;
  if (d->dtype != Y5_F16 || d->KH != 3 || d->KW != 3 || d->SH != 2 || d->SW != 2 || d->PH != 1 || d->PW != 1 || d->C1 != c.c1 || d->Npad != c.nt * 32 ||
      d->C2 > d->Npad || (d->C2 & 7) || oh != d->OH || ow != d->OW || (oh & 3) || (ow & 7) || d->Kpad < 9 * c.c1 || d->H > 255 * 4 || d->W > 65535 || (d->ldx & 7) ||
      d->ldx < d->C1 || !d->act)
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv_k3pw: the 3x3 layer does not match the streaming configuration");
  if (Npad2 != 64 || C3 < 8 || C3 > Npad2 || (C3 & 7) || Kpad2 < d->Npad || (Kpad2 & 7) || (ldy & 7) || split_n < 0 || split_n > C3 || (split_n & 7) ||
      (split_n < C3 && (!y2 || (ld2 & 7) || ld2 < C3 - split_n)) || ldy < (split_n ? split_n : 0) || (split_n == 0 && !y2))
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv_k3pw: the pointwise layer must have <= 64 output channels in multiples of 8, split on a multiple of 8");
  if (((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)bias1 | (uintptr_t)w2_packed | (uintptr_t)bias2 | (uintptr_t)y | (uintptr_t)y2) & 15)
    return y5_fail(Y5_ERR_BAD_ARG, "conv_k3pw: pointers must be 16-byte aligned");
  if ((long long)d->B * d->H * d->W * d->ldx * 2 >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_k3pw: input exceeds 2^31 bytes");
  Y5ConvParams p{};
  p.x = x; p.w = w1_packed; p.bias = bias1; p.y = y; p.y2 = y2;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C1 = d->C1; p.ldx = d->ldx;
  p.OH = oh; p.OW = ow; p.C2 = d->C2; p.ldy = ldy; p.ld2 = ld2;
  p.KH = 3; p.KW = 3; p.SH = 2; p.SW = 2; p.PH = 1; p.PW = 1; p.act = 1;
  p.Kpad = d->Kpad; p.Npad = d->Npad; p.K = 9 * d->C1; p.M = d->B * oh * ow;
  p.x_bytes = (unsigned)((((long long)d->B * d->H * d->W - 1) * d->ldx + d->C1) * 2);
  p.w_bytes = (unsigned)((long long)d->Npad * d->Kpad * 2);
  p.pw2_w = w2_packed; p.pw2_bias = bias2; p.pw2_w_bytes = (unsigned)((long long)Npad2 * Kpad2 * 2);
  p.pw2_kpad = Kpad2; p.pw2_npad = Npad2; p.pw2_c2 = C3; p.pw2_act = act2; p.pw2_split = split_n;
  // two ring stages: with the second filter beside the first a third stage per wave does not fit the 160 KiB (cfg 34 maps to the same kernel)
  if (cfg == 81) return launch_k3_v<32, 2, 2, 1, false, true, 2, false, 8>(p, d->max_blocks, stream);  // eight waves, one stage each
  return launch_k3_v<32, 2, 2, 2, false, true, 2>(p, d->max_blocks, stream);
}

// ---- stem (conv_stem.h) -----------------------------------------------------------------------------------
namespace {
template <int NT, int S, bool RAW = false>
int launch_stem(const Y5StemParams& p, int max_blocks, hipStream_t stream) {
  const size_t lds = y5_conv_stem_lds_bytes<NT, S>();
  auto kern = y5_conv_stem_kernel<NT, S, RAW>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long nbt = ((long long)p.nwt + 3) >> 2;
  long long G = max_blocks;
  if (G <= 0) {
    const int g_num_cu = y5_num_cu();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess || occ < 1) occ = 1;
    G = (long long)g_num_cu * occ;
  }
  if (G > nbt) G = nbt;
  if (G >= 8) G &= ~7LL;
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(256), lds, stream, p);
  return y5_check_launch("y5_conv_stem_fwd");
}
}  // namespace

extern "C" int y5_conv_stem_fwd(const void* x_nchw, int B, int H, int W, const void* w_stem, const float* bias, int C2, int Npad,
                                void* y, int ldy, int max_blocks, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x_nchw || !w_stem || !bias || !y) return y5_fail(Y5_ERR_BAD_ARG, "conv_stem: null pointer");
  if (B < 1 || H < 2 || (H & 1) || W < 64 || (W & 63)) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_stem: needs even H and W % 64 == 0");
  if (C2 < 8 || (C2 & 7) || C2 > Npad || (Npad != 32 && Npad != 64) || (ldy & 7) || ldy < C2)
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv_stem: C2 must be a multiple of 8, <= 64 (Npad 32 or 64)");
  if (((uintptr_t)x_nchw | (uintptr_t)w_stem | (uintptr_t)bias | (uintptr_t)y) & 15) return y5_fail(Y5_ERR_BAD_ARG, "conv_stem: pointers must be 16-byte aligned");
  if ((long long)B * 3 * H * W >= 0x3fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_stem: input exceeds 2^30 elements");
  Y5StemParams p{};
  p.x = x_nchw; p.w = w_stem; p.bias = bias; p.y = y; p.zero = y5_zero_page();
  if (!p.zero) return y5_fail(Y5_ERR_RUNTIME, "conv_stem: zero page allocation failed");
  p.B = B; p.H = H; p.W = W; p.OH = H / 2; p.OW = W / 2; p.C2 = C2; p.ldy = ldy;
  p.tiles_per_row = p.OW / 32;
  p.nwt = B * p.OH * p.tiles_per_row;
  return Npad == 32 ? launch_stem<1, 4>(p, max_blocks, stream) : launch_stem<2, 3>(p, max_blocks, stream);
}

// The same launch without bias and activation: the train-mode forward of 0.Conv (models/common.py:86-88 with BatchNorm in training mode: the
// convolution's own output is what the statistics pass reads).  Replaces a layout pass + a table-gather launch of the general kernel (369 us at bs 64).
extern "C" int y5_conv_stem_fwd_raw(const void* x_nchw, int B, int H, int W, const void* w_stem, int C2, int Npad, void* y, int ldy, int max_blocks,
                                    void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!x_nchw || !w_stem || !y) return y5_fail(Y5_ERR_BAD_ARG, "conv_stem_raw: null pointer");
  if (B < 1 || H < 2 || (H & 1) || W < 64 || (W & 63)) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_stem_raw: needs even H and W % 64 == 0");
  if (C2 < 8 || (C2 & 7) || C2 > Npad || (Npad != 32 && Npad != 64) || (ldy & 7) || ldy < C2)
    return y5_fail(Y5_ERR_UNSUPPORTED, "conv_stem_raw: C2 must be a multiple of 8, <= 64 (Npad 32 or 64)");
  if (((uintptr_t)x_nchw | (uintptr_t)w_stem | (uintptr_t)y) & 15) return y5_fail(Y5_ERR_BAD_ARG, "conv_stem_raw: pointers must be 16-byte aligned");
  if ((long long)B * 3 * H * W >= 0x3fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "conv_stem_raw: input exceeds 2^30 elements");
  Y5StemParams p{};
  p.x = x_nchw; p.w = w_stem; p.y = y; p.zero = y5_zero_page();
  if (!p.zero) return y5_fail(Y5_ERR_RUNTIME, "conv_stem_raw: zero page allocation failed");
  p.bias = static_cast<const float*>(p.zero);   // (the kernel stages Npad floats of bias it never uses)
  p.B = B; p.H = H; p.W = W; p.OH = H / 2; p.OW = W / 2; p.C2 = C2; p.ldy = ldy;
  p.tiles_per_row = p.OW / 32;
  p.nwt = B * p.OH * p.tiles_per_row;
  return Npad == 32 ? launch_stem<1, 4, true>(p, max_blocks, stream) : launch_stem<2, 3, true>(p, max_blocks, stream);
}

#ifdef Y5_K3_TIMING
extern "C" int y5_k3_dbg_read(unsigned long long* out) {  // kernel-experiment builds only (not part of the ABI)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(y5_k3_dbg), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

#ifdef Y5_DBG_TIMING
extern "C" int y5_dbg_read_timing(unsigned long long* out) {  // kernel-experiment builds only (not part of the ABI)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(y5_dbg_timing), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
extern "C" int y5_dbg_read_blocks(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(y5_dbg_blocks), sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : -1;
}
#endif
