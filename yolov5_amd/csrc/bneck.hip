// Fused Bottleneck (conv_bneck.h): y = [x +] SiLU(W2 (*) SiLU(W1 x + b1) + b2) with the 1x1 output kept in LDS -- models/common.py:164-181
// inside C3 (e = 1.0, :242) at the HBM-bound high-resolution levels (C = 32 at P2, 64 at P3 of yolov5s).
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "conv_bneck.h"
#include "conv_h3b.h"
#include "y5_host.h"

namespace {
template <int C, int S, bool ADD, bool CV3 = false, int NWV = 4, bool ALIAS = false>
int launch_bneck(const Y5BneckParams& p, int max_blocks, hipStream_t stream) {
  const size_t lds = y5_conv_bneck_lds_bytes<C, S, CV3, NWV, ALIAS>();
  if (lds > 160 * 1024) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: exceeds 160 KiB of LDS");
  auto kern = y5_conv_bneck_kernel<C, S, ADD, CV3, NWV, ALIAS>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const long long nwt = (long long)p.B * (p.H / 4) * (p.W / 8);
  const long long nbt = (nwt + NWV - 1) / NWV;
  long long G = max_blocks;
  if (G <= 0) {
    const int num_cu = y5_num_cu();
    // resident workgroups per CU: LDS is the limit (52.5 KB at C = 32, one stage: three fit in 160 KB; the occupancy query answers two)
    int occ = (int)((160 * 1024) / lds);
    if (occ < 1) occ = 1;
    if (occ > 4) occ = 4;
    G = (long long)num_cu * occ;
  }
  if (G > nbt) G = nbt;
  if (G >= 8) G &= ~7LL;
  //  the kernel assumes that only a workgroup's LAST tile can be the partial one; with the XCD remap that holds for grids that are a multiple
  // of eight or cover every tile -- an explicit cap of 1..7 below the tile count would put the partial block-tile mid-sequence
  if (G < 8 && G < nbt && nbt >= 8 && nwt % NWV != 0) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: a grid cap below 8 needs a wave-tile count that is a multiple of the waves per workgroup");
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(NWV * 64), lds, stream, p);
  return y5_check_launch("y5_bottleneck_fwd");
}

// ---- c_ = 128: GEMM-1 phase in front of the halo-resident 3x3 (conv_h3b.h) ------------------------------------------------------------------
// spatial tile for an H x W image: TW = ceil(W / d), TH as tall as the 256-pixel MFMA tile and the 320-pixel LDS halo allow, evened out over the
// image height; fewest rounds of `slots` concurrent workgroups wins, ties go to the smaller staged halo (= fewer GEMM-1 rows)
bool h3b_pick_tile(int B, int H, int W, long long slots, int* th, int* tw) {
  constexpr int BM = 256, HPMAX = 320;
  long long best = -1;
  for (int d = 1; d <= 16 && d <= W; ++d) {
    const int TW = (W + d - 1) / d;
    if (TW > BM || TW + 2 > 255) continue;
    int thm = BM / TW < H ? BM / TW : H;
    while (thm >= 1 && (thm + 2) * (TW + 2) > HPMAX) --thm;
    if (thm < 1) continue;
    if (thm + 2 > 255) thm = 253;
    const int nth = (H + thm - 1) / thm;
    const int TH = (H + nth - 1) / nth;
    const int ntw = (W + TW - 1) / TW;
    const long long tiles = (long long)B * nth * ntw;
    const long long rounds = (tiles + slots - 1) / slots;
    const long long cost = rounds * (1LL << 32) + (long long)(TH + 2) * (TW + 2) * nth * ntw;
    if (best < 0 || cost < best) { best = cost; *th = TH; *tw = TW; }
  }
  return best >= 0;
}

template <int NSW, bool K64 = false, bool CV3 = false>
int launch_h3b(Y5H3bParams p, int max_blocks, hipStream_t stream) {
  using Gm = Y5H3bGeom<NSW, K64, CV3>;
  auto kern = y5_conv_h3b_kernel<NSW, K64, CV3>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  long long G = max_blocks;
  if (G <= 0) {
    const int num_cu = y5_num_cu();
    G = num_cu;   // 145 KB of LDS: one workgroup per CU
  }
  if (!h3b_pick_tile(p.B, p.H, p.W, G, &p.th, &p.tw)) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: no spatial tile fits the 320-pixel halo");
  p.tiles_h = (p.H + p.th - 1) / p.th;
  p.tiles_w = (p.W + p.tw - 1) / p.tw;
  const long long ntiles = (long long)p.B * p.tiles_h * p.tiles_w;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return y5_fail(Y5_ERR_BAD_ARG, "bottleneck: grid out of range");
  if (G > ntiles) G = ntiles;
  if (G >= 8) G &= ~7LL;
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(Gm::NW * 64), Gm::LDS, stream, p);
  return y5_check_launch("y5_bottleneck_fwd(h3b)");
}
}  // namespace

extern "C" int y5_bottleneck_fwd(const void* x, int ldx, const void* w1_packed, const float* bias1, int Kpad1, const void* w2_packed,
                                 const float* bias2, int Kpad2, void* y, int ldy, int B, int H, int W, int C, int add, int max_blocks,
                                 void* stream_) {
  if (!x || !w1_packed || !bias1 || !w2_packed || !bias2 || !y) return y5_fail(Y5_ERR_BAD_ARG, "bottleneck: null pointer");
  if (C != 32 && C != 64 && C != 128) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: fused kernel exists for 32, 64 and 128 channels");
  if (C != 128 && (B < 1 || H < 4 || W < 8 || (H & 3) || (W & 7) || H > 255 * 4 || W > 65535))
    return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: needs H % 4 == 0 and W % 8 == 0");
  if (C == 128 && (B < 1 || H < 1 || W < 1 || H > 65535 || W > 65535)) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: bad image size");
  if ((ldx & 7) || (ldy & 7) || ldx < C || ldy < C || Kpad1 < C || (Kpad1 & 7) || Kpad2 < 9 * C || (Kpad2 & 7))
    return y5_fail(Y5_ERR_BAD_ARG, "bottleneck: bad strides / packed filter dims");
  if (((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)bias1 | (uintptr_t)w2_packed | (uintptr_t)bias2 | (uintptr_t)y) & 15)
    return y5_fail(Y5_ERR_BAD_ARG, "bottleneck: pointers must be 16-byte aligned");
  const long long npix = (long long)B * H * W;
  if (npix * ldx * 2 >= 0x7fffffffLL || npix * ldy * 2 >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: tensor exceeds 2^31 bytes");
  const char* xb = static_cast<const char*>(x);
  char* yb = static_cast<char*>(y);
  if (xb < yb + ((npix - 1) * ldy + C) * 2 && yb < xb + ((npix - 1) * ldx + C) * 2 && (ldx != ldy || ((yb - xb) % (ldx * 2) + ldx * 2) % (ldx * 2) < C * 2 || ((xb - yb) % (ldx * 2) + ldx * 2) % (ldx * 2) < C * 2))
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
  if (C == 128) {   // conv_h3b.h: 1x1 as a GEMM-1 phase of the halo-resident 3x3; stages = depth of the 3x3 filter ring (9 default; 18 = 9 stages with double steps; 4, 5)
    Y5H3bParams q{};
    q.x = x; q.w1 = w1_packed; q.w2 = w2_packed; q.b1 = bias1; q.b2 = bias2; q.y = y;
    q.x_bytes = p.x_bytes; q.w1_bytes = p.w1_bytes; q.w2_bytes = p.w2_bytes;
    q.B = B; q.H = H; q.W = W; q.ldx = ldx; q.ldy = ldy; q.Kpad1 = Kpad1; q.Kpad2 = Kpad2; q.add = add;
    if (stages == 0 || stages == 9) return launch_h3b<9>(q, max_blocks, st);      // W1 streamed through the ring, one barrier per slice
    if (stages == 18) return launch_h3b<9, true>(q, max_blocks, st);               // ... one barrier per TWO slices (profiles/r05/r05_h3b_v3_k64.log: not faster)
    if (stages == 4) return launch_h3b<4>(q, max_blocks, st);                    // W1 resident behind a short ring
    if (stages == 5) return launch_h3b<5>(q, max_blocks, st);
    return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: unsupported number of stages");
  }
  const int S = stages > 0 ? stages : 1;  // measured (scripts/bneck_bench.py): resident waves beat prefetch depth -- 1 stage, 3 workgroups per CU
  if (C == 32) {
    if (S == 1) return add ? launch_bneck<32, 1, true>(p, max_blocks, st) : launch_bneck<32, 1, false>(p, max_blocks, st);
    if (S == 2) return add ? launch_bneck<32, 2, true>(p, max_blocks, st) : launch_bneck<32, 2, false>(p, max_blocks, st);
    if (S == 3) return add ? launch_bneck<32, 3, true>(p, max_blocks, st) : launch_bneck<32, 3, false>(p, max_blocks, st);
  } else if (stages == 0) {   // C = 64 default (round 4): eight waves, t aliased onto the single stage
    return add ? launch_bneck<64, 1, true, false, 8, true>(p, max_blocks, st) : launch_bneck<64, 1, false, false, 8, true>(p, max_blocks, st);
  } else if (S == 1) {        // stages = 1 asked for explicitly: the four-wave form with stage and t apart
    return add ? launch_bneck<64, 1, true>(p, max_blocks, st) : launch_bneck<64, 1, false>(p, max_blocks, st);
  }
  return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck: unsupported number of stages");
}


// Bottleneck + C3's cv3 as one launch (conv_bneck.h CV3): out = act3(W3 [y ; y2] + b3) with y = the Bottleneck's result (never stored)
extern "C" int y5_bottleneck_cv3_fwd(const void* x, int ldx, const void* w1_packed, const float* bias1, int Kpad1, const void* w2_packed,
                                     const float* bias2, int Kpad2, const void* y2, int ld2, const void* w3_packed, const float* bias3, int Kpad3,
                                     int C3, int act3, void* out, int ldo, int B, int H, int W, int C, int add, int max_blocks, void* stream_) {
  if (!x || !w1_packed || !bias1 || !w2_packed || !bias2 || !y2 || !w3_packed || !bias3 || !out) return y5_fail(Y5_ERR_BAD_ARG, "bottleneck_cv3: null pointer");
  if (C != 32 && C != 128) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck_cv3: built for 32-channel (cv3: 64 -> <= 64) and 128-channel (cv3: 256 -> <= 256) Bottlenecks");
  if (C == 32 && (B < 1 || H < 4 || W < 8 || (H & 3) || (W & 7) || H > 255 * 4 || W > 65535)) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck_cv3: needs H % 4 == 0 and W % 8 == 0");
  if (C == 128 && (B < 1 || H < 1 || W < 1 || H > 65535 || W > 65535)) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck_cv3: bad image size");
  if ((ldx & 7) || (ld2 & 7) || (ldo & 7) || ldx < C || ld2 < C || Kpad1 < C || (Kpad1 & 7) || Kpad2 < 9 * C || (Kpad2 & 7) || Kpad3 < 2 * C || (Kpad3 & 7) ||
      C3 < 8 || C3 > 2 * C || (C3 & 7) || ldo < C3)
    return y5_fail(Y5_ERR_BAD_ARG, "bottleneck_cv3: bad strides / packed filter dims / output channels");
  if (((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)bias1 | (uintptr_t)w2_packed | (uintptr_t)bias2 | (uintptr_t)y2 | (uintptr_t)w3_packed | (uintptr_t)bias3 |
       (uintptr_t)out) & 15)
    return y5_fail(Y5_ERR_BAD_ARG, "bottleneck_cv3: pointers must be 16-byte aligned");
  const long long npix = (long long)B * H * W;
  if (npix * ldx * 2 >= 0x7fffffffLL || npix * ld2 * 2 >= 0x7fffffffLL || npix * ldo * 2 >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "bottleneck_cv3: tensor exceeds 2^31 bytes");
  auto overlaps = [&](const void* a_, long long lda, int ca, const void* b_, long long ldb, int cb) {  // channel slices of NHWC buffers
    const char* a = static_cast<const char*>(a_);
    const char* b = static_cast<const char*>(b_);
    if (!(a < b + ((npix - 1) * ldb + cb) * 2 && b < a + ((npix - 1) * lda + ca) * 2)) return false;  // exact extents: a slice ends ca channels into its last pixel
    if (lda != ldb) return true;
    const long long d = ((b - a) % (lda * 2) + lda * 2) % (lda * 2), e = ((a - b) % (lda * 2) + lda * 2) % (lda * 2);
    return d < ca * 2 || e < cb * 2;
  };
  if (overlaps(x, ldx, C, out, ldo, C3) || overlaps(y2, ld2, C, out, ldo, C3))
    return y5_fail(Y5_ERR_BAD_ARG, "bottleneck_cv3: out must not overlap x or y2 (tiles read their neighbours' pixels of x)");
  Y5BneckParams p{};
  p.x = x; p.w1 = w1_packed; p.w2 = w2_packed; p.b1 = bias1; p.b2 = bias2; p.y = out;
  p.B = B; p.H = H; p.W = W; p.ldx = ldx; p.ldy = ldo; p.Kpad1 = Kpad1; p.Kpad2 = Kpad2; p.add = add;
  p.x_bytes = (unsigned)(((npix - 1) * ldx + C) * 2);
  p.w1_bytes = (unsigned)((long long)C * Kpad1 * 2);
  p.w2_bytes = (unsigned)((long long)C * Kpad2 * 2);
  p.y2 = y2; p.ld2 = ld2; p.y2_bytes = (unsigned)(((npix - 1) * ld2 + C) * 2);
  p.w3 = w3_packed; p.b3 = bias3; p.Kpad3 = Kpad3; p.C3 = C3; p.act3 = act3; p.w3_bytes = (unsigned)((long long)2 * C * Kpad3 * 2);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  max_blocks &= 0xffff;
  if (C == 128) {   // conv_h3b.h CV3: GEMM-1 phase + halo-resident 3x3 + cv3 as a GEMM-3 phase (the Bottleneck's result stays in LDS)
    Y5H3bParams q{};
    q.x = x; q.w1 = w1_packed; q.w2 = w2_packed; q.b1 = bias1; q.b2 = bias2; q.y = out;
    q.x_bytes = p.x_bytes; q.w1_bytes = p.w1_bytes; q.w2_bytes = p.w2_bytes;
    q.B = B; q.H = H; q.W = W; q.ldx = ldx; q.ldy = ldo; q.Kpad1 = Kpad1; q.Kpad2 = Kpad2; q.add = add;
    q.y2 = y2; q.ld2 = ld2; q.y2_bytes = p.y2_bytes; q.w3 = w3_packed; q.b3 = bias3; q.Kpad3 = Kpad3; q.C3 = C3; q.act3 = act3;
    q.w3_bytes = (unsigned)((long long)((C3 + 31) / 32 * 32) * Kpad3 * 2);
    return launch_h3b<9, false, true>(q, max_blocks, st);
  }
  return add ? launch_bneck<32, 1, true, true>(p, max_blocks, st) : launch_bneck<32, 1, false, true>(p, max_blocks, st);
}

#ifdef Y5_H3B_TIMING
extern "C" int y5_h3b_dbg_read(unsigned long long* out) {  // kernel-experiment builds only (not part of the ABI)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(y5_h3b_stamps), sizeof(unsigned long long) * 512 * 4 * 8) == hipSuccess ? 0 : -1;
}
#endif
