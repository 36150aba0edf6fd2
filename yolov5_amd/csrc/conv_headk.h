// Detect convolution + Detect decode in ONE launch for the DEEP pyramid levels (models/yolo.py:91-115: x[i] = self.m[i](x[i]) ... y = x[i].sigmoid(); xy / wh
// decode; z.append(y.view(bs, -1, no))) -- P4 (256 input channels) and P5 (512) of yolov5s.  The P3 level runs conv_pw.h's fused head (filter resident in
// LDS); with K = 256 / 512 the 255 x K filter no longer fits beside the per-wave rings, so here K STREAMS: per 32-channel chunk the 256-pixel x tile
// (16 KB) and the filter chunk (256 rows x 64 B = 16 KB) go through an NS-stage LDS ring (LDS-DMA, counted vmcnt, one barrier per chunk), every wave
// multiplies its own 32 pixels against all 256 output channels (8 accumulator blocks), and the logits never reach HBM: the epilogue is conv_pw.h's DEC
// epilogue -- per anchor the wave parks its 32 x 85 values (fp16-rounded logits for the box outputs, sigmoid otherwise) in LDS, 32 lanes turn the four box
// logits of their pixel into xy / wh, and the block leaves as 16-byte stores into the anchor's z rows -- bit-identical to y5_conv2d_fwd(act = 0) +
// y5_detect_decode (same fp16-rounded logits, same arithmetic; head.hip is built with -ffp-contract=off like the decode kernel).
// A wave's 32 pixels may straddle two images (P5: 400 pixels per image): rows are addressed per image, the z block is stored as two segments whose
// boundary is a multiple of eight rows = 85 16-byte vectors.
#pragma once
#include "conv_pw.h"   // Y5HeadParams

template <int NS_>
struct Y5HeadkGeom {
  static constexpr int NW = 8, NS = NS_, BM = NW * 32, NT = 8, NPAD = NT * 32, NO = 85;
  static constexpr int XS = BM * 64, WS = NPAD * 64, STAGE = XS + WS;
  static constexpr int XPW = BM / 16 / NW, WPW = NPAD / 16 / NW, PPW = XPW + WPW;   // LDS-DMA instructions per wave per chunk
  static constexpr int PARK = 32 * NO * 2;                                            // per-wave scratch of the epilogue (inside the idle ring)
  static constexpr size_t RING = (size_t)NS * STAGE, OFF_BIAS = RING, OFF_DUMMY = OFF_BIAS + NPAD * 4, LDS = OFF_DUMMY + 1024;
  static_assert((size_t)NW * PARK <= RING && LDS <= 160 * 1024 && NS >= 3 && PARK % 16 == 0, "LDS budget");
};

template <int NS_, bool HINT>
__global__ __launch_bounds__(512, 2)
void y5_conv_headk_kernel(const Y5ConvParams p, const Y5HeadParams hd) {
  using Gm = Y5HeadkGeom<NS_>;
  constexpr int NW = Gm::NW, NS = Gm::NS, NT = Gm::NT, NO = Gm::NO, STAGE = Gm::STAGE, XS = Gm::XS, XPW = Gm::XPW, WPW = Gm::WPW, PPW = Gm::PPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const blds = reinterpret_cast<float*>(smem + Gm::OFF_BIAS);
  char* const dummy = smem + Gm::OFF_DUMMY;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, frow = lane & 31;
  const int NK = p.C1 >> 5;
  const int m00 = blockIdx.x * Gm::BM;           // first pixel of the workgroup tile (pixels are rows of the (B npix) x C1 matrix)
  const int m0 = m00 + wave * 32;                // ... of this wave's 32
  const bool live = m0 < p.M;                    // (M % 32 == 0: a wave tile is all real pixels or none)

  const y5_rsrc_t xrs = y5_make_rsrc(p.x, p.x_bytes);
  const y5_rsrc_t wrs = y5_make_rsrc(p.w, p.w_bytes);
  for (int i = tid; i < Gm::NPAD; i += NW * 64) blds[i] = p.bias[i];

  unsigned x_off[XPW], w_off[WPW];
#pragma unroll
  for (int q = 0; q < XPW; ++q) {
    const int idx = (q * NW + wave) * 64 + lane;
    const int row = idx >> 2, ss = (idx & 3) ^ ((row >> 2) & 3);
    x_off[q] = m00 + row < p.M ? (unsigned)(((long long)(m00 + row) * p.ldx) * 2 + ss * 16) : Y5_OOB;
  }
#pragma unroll
  for (int q = 0; q < WPW; ++q) {
    const int idx = (q * NW + wave) * 64 + lane;
    const int row = idx >> 2, ss = (idx & 3) ^ ((row >> 2) & 3);
    w_off[q] = (unsigned)((row * p.Kpad) * 2 + ss * 16);
  }
  auto issue = [&](int c) {
    char* st = smem + (c % NS) * STAGE;
#pragma unroll
    for (int q = 0; q < XPW; ++q) y5_bglds16(xrs, x_off[q] == Y5_OOB ? Y5_OOB : x_off[q] + (unsigned)(c * 64), st + (q * NW + wave) * 1024);
#pragma unroll
    for (int q = 0; q < WPW; ++q) y5_bglds16(wrs, w_off[q] + (unsigned)(c * 64), st + XS + (q * NW + wave) * 1024);
  };
  auto issue_dummy = [&]() {
#pragma unroll
    for (int q = 0; q < PPW; ++q) y5_bglds16_dummy(xrs, dummy);
  };

  const int hp = wave * 32 + frow;
  const int a_rd = (hp << 6) | ((g ^ ((hp >> 2) & 3)) << 4);
  const int w_rd = XS + (frow << 6) + ((g ^ ((frow >> 2) & 3)) << 4);   // + j * 32 rows
  float16_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#pragma unroll
  for (int c = 0; c < NS - 1; ++c)
    if (c < NK) issue(c);
    else issue_dummy();
  for (int c = 0; c < NK; ++c) {
    y5_wait_vm<(NS - 2) * PPW>();   // chunk c has landed (chunks c+1 .. c+NS-2 may be in flight; dummies past the end keep the count constant)
    __builtin_amdgcn_s_barrier();
    if (c + NS - 1 < NK) issue(c + NS - 1);   // into the stage chunk c - 1 occupied: every wave finished reading it before this barrier
    else issue_dummy();
    const char* st = smem + (c % NS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const half8_t af = *reinterpret_cast<const half8_t*>(st + (a_rd ^ (ks * 32)));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const half8_t wf = *reinterpret_cast<const half8_t*>(st + ((w_rd + j * 32 * 64) ^ (ks * 32)));
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc[j], 0, 0, 0);
      }
    }
  }
  y5_wait_vm<0>();
  __syncthreads();   // every wave is done with the ring: a wave's slice of it becomes its decode scratch
  if (!live) return;

  // ---- epilogue (conv_pw.h DEC): per anchor, 32 x 85 values through the wave's scratch -> decoded z rows ------------------------------------------------
  half_t* sc = reinterpret_cast<half_t*>(smem + wave * Gm::PARK);
  const int bimg = m0 / hd.npix;
  const int pix0 = m0 - bimg * hd.npix;
  const int rb = hd.npix - pix0 < 32 ? hd.npix - pix0 : 32;   // rows of this wave tile that belong to image bimg (a multiple of 8); the rest open image bimg + 1
  const int lb = lane < rb ? bimg : bimg + 1;                 // this lane's pixel (lanes 0..31)
  const int lpix = lane < rb ? pix0 + lane : lane - rb;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j * 32 + 31 < NO * a || j * 32 >= NO * a + NO) continue;  // sub-tile outside this anchor's channels
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = j * 32 + q * 8;  // this lane's channels: n0 + g*4 + e
        if (n0 + 7 < NO * a || n0 >= NO * a + NO) continue;
        const float4_t bv = *reinterpret_cast<const float4_t*>(blds + j * 32 + q * 8 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int o = n0 + g * 4 + e - NO * a;
          const half_t v16 = (half_t)(acc[j][q * 4 + e] + bv[e]);   // the logit as the unfused path stores it
          const float v = (float)v16;
          const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
          const half_t val = o < 4 ? v16 : (half_t)sg;
          if (o >= 0 && o < NO) sc[frow * NO + o] = val;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) {
      const int iy = (int)__umulhi((unsigned)lpix, hd.inv_nx), ix = lpix - iy * hd.nx;
      half_t* qv = sc + lane * NO;
      const float gx = (float)ix - 0.5f, gy = (float)iy - 0.5f;
      const float aw = hd.anchors_px[a * 2], ah = hd.anchors_px[a * 2 + 1];
      float s2[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) s2[o] = __builtin_amdgcn_rcpf(1.0f + __expf(-(float)qv[o])) * 2.0f;
      qv[0] = (half_t)((s2[0] + gx) * hd.stride);   // yolo.py:110
      qv[1] = (half_t)((s2[1] + gy) * hd.stride);
      qv[2] = (half_t)(s2[2] * s2[2] * aw);         // yolo.py:111
      qv[3] = (half_t)(s2[3] * s2[3] * ah);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    half_t* z0 = static_cast<half_t*>(hd.z) + ((long long)bimg * hd.nrows_total + hd.row_off + (long long)a * hd.npix + pix0) * NO;
    half_t* z1 = static_cast<half_t*>(hd.z) + ((long long)(bimg + 1) * hd.nrows_total + hd.row_off + (long long)a * hd.npix) * NO - (long long)rb * NO;
    constexpr int NV = 32 * NO / 8;  // 16-byte vectors per anchor block
    const int vb = rb * NO / 8;      // first vector of the second segment (rb % 8 == 0)
#pragma unroll
    for (int it = 0; it < (NV + 63) / 64; ++it) {
      const int v = it * 64 + lane;
      if (v < NV) *reinterpret_cast<uint4_t*>((v < vb ? z0 : z1) + v * 8) = *reinterpret_cast<const uint4_t*>(sc + v * 8);
    }
    if constexpr (HINT) {  // the rows' objectness, bit for bit what z holds
      if (lane < 32)
        static_cast<half_t*>(hd.obj_hint)[(long long)lb * hd.nrows_total + hd.row_off + (long long)a * hd.npix + lpix] = sc[lane * NO + 4];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the scratch is rewritten by the next anchor
  }
}
