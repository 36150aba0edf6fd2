// SPPF front half as ONE launch (models/common.py:318-340): x -> cv1 (1x1, BN folded, SiLU) -> three cascaded k x k stride-1 max pools, the four results written
// to the four channel slices of the concat buffer that cv2 reads (y = cat(x1, y1, y2, y3), :340).  Two launches (the cv1 GEMM on 128 x 128 tiles over
// all pixels, then y5_sppf_pool_kernel re-reading x1 per image) become one because a workgroup here owns what the pools need: ALL H x W pixels of one
// image for a group of 64 output channels.
//   * GEMM: M = H W <= 416 pixels (13 MFMA row blocks), N = 64, K = C1, streamed in 32-channel chunks through an NS-stage LDS ring (x chunk: H W x 64 B,
//     filter chunk: 64 x 64 B; LDS-DMA, counted vmcnt, one barrier per chunk); the 26 accumulator blocks are dealt over eight waves (<= 4 each: channel
//     block w & 1, row blocks (w >> 1) + 4 i).  The image's x is read by the c_ / 64 workgroups of its channel groups (L2 hits after the first).
//   * epilogue: bias + SiLU -> fp16 plane y0 [H W][64 ch] in LDS (row pitch 144 B: the 8-byte accumulator stores of 32 pixels do not pile onto one bank)
//     -> slice 0 as 16-byte row-contiguous stores -- exactly the bytes the unfused cv1 stores;
//   * pools: per 32-channel half, separable row / column max passes on dense 64-byte-pitch planes (tmp, two ping-pong outputs), per-thread windows
//     precomputed once (misc_kernels.h y5_sppf_pool_kernel's scheme); every pass's output leaves as 64 contiguous bytes per pixel into its slice.
// LDS: max(ring NS x 30 KB, y0 58.5 KB + three 25.6 KB planes) + bias = 136 KB: one workgroup of eight waves per CU; grid = B x c_ / 64.
#pragma once
#include "conv_igemm.h"

struct Y5SppfParams {
  const void* x;        // (B, H, W, C1) NHWC slice, pixel stride ldx
  const void* w;        // [c_ padded to 64][Kpad] fp16, k = c
  const float* bias;    // [c_]
  void* buf;            // concat buffer: slice s (s = 0..3) = channels [s c_, (s + 1) c_), pixel stride ld
  unsigned x_bytes, w_bytes;
  int B, H, W, C1, ldx, c_, ld, Kpad, k, act;
};

template <int NS_>
struct Y5SppfGeom {
  static constexpr int NW = 8, NS = NS_, MAXHW = 416, NRB = MAXHW / 32, NBLK = 4;   // row blocks per wave: (w >> 1) + 4 i
  static constexpr int NAIX = MAXHW / 16;                   // x pieces (16 pixels x 64 B) per chunk
  static constexpr int XS = NAIX * 1024, WS = 64 * 64, STAGE = XS + WS;
  static constexpr int XPW = (NAIX + NW - 1) / NW, PPW = XPW + 1;   // LDS-DMA instructions per wave per chunk (dummies keep the count uniform)
  static constexpr int Y0_PITCH = 144, PLANE = MAXHW * 64;  // y0: 128 B of channels + 16 B pad; dense half planes: 32 channels per pixel
  static constexpr size_t RING = (size_t)NS * STAGE, POOL = (size_t)MAXHW * Y0_PITCH + 3 * (size_t)PLANE;
  static constexpr size_t OFF_BIAS = RING > POOL ? RING : POOL, OFF_DUMMY = OFF_BIAS + 256, LDS = OFF_DUMMY + 1024;
  static_assert(LDS <= 160 * 1024 && NS >= 3, "LDS budget");
};

template <int NS_>
__global__ __launch_bounds__(512, 2)
void y5_sppf_cv1_pool_kernel(const Y5SppfParams p) {
  typedef half_t T;
  typedef half8_t V;
  using Gm = Y5SppfGeom<NS_>;
  constexpr int NW = Gm::NW, NS = Gm::NS, NBLK = Gm::NBLK, STAGE = Gm::STAGE, XS = Gm::XS, XPW = Gm::XPW, PPW = Gm::PPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const bias_lds = reinterpret_cast<float*>(smem + Gm::OFF_BIAS);
  char* const dummy = smem + Gm::OFF_DUMMY;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, frow = lane & 31;
  const int HW = p.H * p.W;
  const int groups = p.c_ >> 6;
  const int b = blockIdx.x / groups, cg = blockIdx.x - b * groups;
  const int NK = p.C1 >> 5;
  const int NAI = (HW + 15) >> 4;

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t wrs = y5_make_rsrc(p.w, p.w_bytes);
  if (tid < 64) bias_lds[tid] = p.bias[cg * 64 + tid];

  // ---- loader: per chunk a wave issues XPW pieces of x (piece I = q NW + wave: 16 pixels x 4 slots, source-side swizzle) and one filter piece -----------
  unsigned x_off[XPW];
#pragma unroll
  for (int q = 0; q < XPW; ++q) {
    const int idx = (q * NW + wave) * 64 + lane;
    const int hp = idx >> 2, ss = (idx & 3) ^ ((hp >> 2) & 3);
    x_off[q] = (q * NW + wave) < NAI && hp < HW ? (unsigned)((((long long)b * HW + hp) * p.ldx) * 2 + ss * 16) : Y5_OOB;
  }
  unsigned w_off;
  {
    const int row = wave * 16 + (lane >> 2);
    const int ss = (lane & 3) ^ ((row >> 2) & 3);
    w_off = wave < 4 ? (unsigned)(((cg * 64 + row) * p.Kpad) * 2 + ss * 16) : Y5_OOB;
  }
  auto issue = [&](int c) {
    char* st = smem + (c % NS) * STAGE;
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
      const int I = q * NW + wave;
      if (I < Gm::NAIX) y5_bglds16(xrs, x_off[q] == Y5_OOB ? Y5_OOB : x_off[q] + (unsigned)(c * 64), st + I * 1024);
      else y5_bglds16_dummy(xrs, dummy);
    }
    if (wave < 4) y5_bglds16(wrs, w_off + (unsigned)(c * 64), st + XS + wave * 1024);
    else y5_bglds16_dummy(wrs, dummy);
  };

  // ---- GEMM ----------------------------------------------------------------------------------------------------------------------------------------
  const int cb = wave & 1, rb0 = wave >> 1;
  const int w_rd = (cb * 32 + frow) * 64 + ((g ^ ((frow >> 2) & 3)) << 4);
  int a_rd[NBLK];
#pragma unroll
  for (int i = 0; i < NBLK; ++i) {
    const int hp = (rb0 + 4 * i) * 32 + frow;   // (rb0 + 12 = 13..15 for waves 2..7 lies beyond the 13 row blocks: multiplied, never stored; reads stay inside the stage)
    a_rd[i] = ((hp < Gm::MAXHW ? hp : 0) << 6) | ((g ^ ((hp >> 2) & 3)) << 4);
  }
  float16_t acc[NBLK];
#pragma unroll
  for (int i = 0; i < NBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

#pragma unroll
  for (int c = 0; c < NS - 1; ++c)
    if (c < NK) issue(c);
    else { for (int q = 0; q < PPW; ++q) y5_bglds16_dummy(xrs, dummy); }
  for (int c = 0; c < NK; ++c) {
    // chunk c has landed: everything but the chunks issued after it (c+1 .. c+NS-2, PPW instructions each; dummies past the end keep this a constant)
    y5_wait_vm<(NS - 2) * PPW>();
    __builtin_amdgcn_s_barrier();
    // chunk c + NS - 1 goes into the stage chunk c - 1 occupied (every wave finished reading it before this barrier)
    if (c + NS - 1 < NK) issue(c + NS - 1);
    else { for (int q = 0; q < PPW; ++q) y5_bglds16_dummy(xrs, dummy); }
    const char* st = smem + (c % NS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const half8_t wf = *reinterpret_cast<const half8_t*>(st + XS + (w_rd ^ (ks * 32)));
      half8_t af[NBLK];
#pragma unroll
      for (int i = 0; i < NBLK; ++i) af[i] = *reinterpret_cast<const half8_t*>(st + (a_rd[i] ^ (ks * 32)));
#pragma unroll
      for (int i = 0; i < NBLK; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af[i], acc[i], 0, 0, 0);
    }
  }
  y5_wait_vm<0>();     // (only dummies are left in flight)
  __syncthreads();     // every wave is done with the ring: it becomes the pooling planes

  // ---- epilogue: bias + SiLU -> y0 [pixel][64 channels], pitch 144 B ------------------------------------------------------------------------------------
  char* const y0 = smem;
  V* const tmp = reinterpret_cast<V*>(smem + (size_t)Gm::MAXHW * Gm::Y0_PITCH);
  V* const pa = tmp + Gm::PLANE / 16;
  V* const pb = pa + Gm::PLANE / 16;
#pragma unroll
  for (int i = 0; i < NBLK; ++i) {
    const int hp = (rb0 + 4 * i) * 32 + frow;
    if (rb0 + 4 * i < Gm::NRB && hp < HW) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4_t bq = *reinterpret_cast<const float4_t*>(bias_lds + cb * 32 + q * 8 + g * 4);
        uint2_t o;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = acc[i][q * 4 + e] + bq[e]; v[e] = p.act ? y5_silu(t) : t; }
        o[0] = y5_pack_h2(v[0], v[1]);
        o[1] = y5_pack_h2(v[2], v[3]);
        *reinterpret_cast<uint2_t*>(y0 + hp * Gm::Y0_PITCH + (cb * 32 + q * 8 + g * 4) * 2) = o;
      }
    }
  }
  __syncthreads();
  char* const base = static_cast<char*>(p.buf) + ((size_t)b * HW * p.ld + (size_t)cg * 64) * 2;   // this group's 64 channels of slice 0
  const size_t ldb = (size_t)p.ld * 2, slice = (size_t)p.c_ * 2;
  for (int v = tid; v < HW * 8; v += NW * 64) {   // slice 0 = cv1's output: 128 contiguous bytes per pixel
    const int i = v >> 3, j = v & 7;
    *reinterpret_cast<uint4_t*>(base + (size_t)i * ldb + j * 16) = *reinterpret_cast<const uint4_t*>(y0 + i * Gm::Y0_PITCH + j * 16);
  }

  // ---- three cascaded k x k max pools per 32-channel half (max is exact: separable row pass into tmp, column pass out of it) ------------------------------
  constexpr int GV = 4, MAXIT = (Gm::MAXHW * GV + NW * 64 - 1) / (NW * 64);
  const int n = HW * GV, r = p.k / 2, W = p.W, H = p.H, cs = W * GV;
  int rowlo[MAXIT], rowcnt[MAXIT], collo[MAXIT], colcnt[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int v = tid + it * NW * 64;
    const int vv = v < n ? v : 0;
    const int i = vv / GV, gl = vv % GV;
    const int y = i / W, x = i - y * W;
    const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
    const int yy0 = y - r < 0 ? 0 : y - r, yy1 = y + r >= H ? H - 1 : y + r;
    rowlo[it] = (y * W + x0) * GV + gl; rowcnt[it] = v < n ? x1 - x0 : -1;   // window = first element + cnt more, stride GV (pixels of one row)
    collo[it] = (yy0 * W + x) * GV + gl; colcnt[it] = yy1 - yy0;            // ... stride W GV (rows of one column)
  }
  for (int hf = 0; hf < 2; ++hf) {
    V* in = nullptr;   // pass 1 reads the pitched y0 plane
    V* out = pa;
    for (int pass = 1; pass <= 3; ++pass) {
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (rowcnt[it] >= 0) {
          V m;
          if (pass == 1) {
            const int i0 = rowlo[it] / GV, gl = rowlo[it] % GV;
            const char* q = y0 + i0 * Gm::Y0_PITCH + hf * 64 + gl * 16;
            m = *reinterpret_cast<const V*>(q);
            for (int j = 1; j <= rowcnt[it]; ++j) m = __builtin_elementwise_max(m, *reinterpret_cast<const V*>(q + j * Gm::Y0_PITCH));
          } else {
            const V* q = in + rowlo[it];
            m = q[0];
            for (int j = 1; j <= rowcnt[it]; ++j) m = __builtin_elementwise_max(m, q[j * GV]);
          }
          tmp[tid + it * NW * 64] = m;
        }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (rowcnt[it] >= 0) {
          const int v = tid + it * NW * 64;
          const V* q = tmp + collo[it];
          V m = q[0];
          for (int j = 1; j <= colcnt[it]; ++j) m = __builtin_elementwise_max(m, q[j * cs]);
          out[v] = m;
          *reinterpret_cast<V*>(base + (size_t)(v / GV) * ldb + (size_t)pass * slice + hf * 64 + (v % GV) * 16) = m;
        }
      __syncthreads();
      in = out;
      out = out == pa ? pb : pa;
    }
  }
}
