// Pointwise (1x1, stride 1) convolution for the HBM-bound high-resolution layers of the YOLOv5 graph
// (C3 cv1+cv2 / cv3 / Bottleneck.cv1 at P1..P3: AI 16..100 flop/B, far below the MFMA ridge).  Same math and the same
// operand layouts as conv_igemm.h (D[n][m] = sum_k W[n][k] * A[m][k], mfma_f32_32x32x16_f16, XOR-swizzled LDS rows
// filled by LDS-DMA), but organised as a STREAMING kernel:
//   * the whole filter [Npad][K] and the bias live in LDS for the lifetime of the persistent workgroup;
//   * every wave owns a private ring of S stages, one stage = 32 consecutive pixels x K channels, filled by its own
//     global_load_lds instructions and retired with COUNTED s_waitcnt vmcnt(N) -- the queue is never drained in steady
//     state, S-1 tiles of loads plus the previous tiles' stores stay in flight per wave;
//   * waves never synchronise with each other after the prologue (no s_barrier in the loop);
//   * the epilogue (bias + SiLU) is transposed through the stage buffer the tile just vacated and leaves as full-row
//     16-byte stores (plus the optional 2x nearest-upsampled replica of nn.Upsample, yolov5s.yaml:36,41).
// vmcnt bookkeeping (gfx9 family: loads and stores retire in issue order on one counter): with LP loads and SP stores
// per tile, the operations issued after tile i's loads are (S-1) x (LP + SP) in steady state and (S-1) x LP + i x SP
// while the ring is filling; when fewer than S-1 tiles remain the wave drains completely.
#pragma once
#include "conv_igemm.h"

template <int KC, int RB, int NT, int S, int OS = 1, int NWV = 4>
constexpr size_t y5_conv_pw_lds_bytes() {
  constexpr int NPAD = 32 * NT;
  constexpr int STAGE_A = 32 * KC * RB, STAGE_O = 32 * NPAD * 2 / OS;
  constexpr int STAGE = STAGE_A > STAGE_O ? STAGE_A : STAGE_O;
  return (size_t)KC * NPAD * RB + (size_t)NPAD * 4 + (size_t)NWV * S * STAGE;
}

// OS > 1 (wide outputs, e.g. the 255-channel Detect heads): the epilogue leaves in OS channel groups of NT/OS sub-tiles each, so
// that the transposition scratch is not larger than an input stage and filter + rings still fit the 160 KB of LDS.
// Parameters of the fused Detect-head epilogue (DEC): the tile's logits never reach HBM; they are decoded
// (models/yolo.py:102-111, same arithmetic as y5_detect_decode_kernel's z-only path) and leave as the z rows of the three anchors.
struct Y5HeadParams {
  void* z;                 // (B, nrows_total, 85) fp16
  long long nrows_total, row_off;
  int npix, nx;            // pixels per image (multiple of 32), grid width
  unsigned inv_nx;         // ceil(2^32 / nx)
  float stride;
  float anchors_px[6];     // 3 anchors x (w, h) in pixels
  void* obj_hint;          // HINT kernels: (B, nrows_total) fp16 plane receiving a copy of every row's objectness (z[..., 4]) for the NMS filter
};

template <int KC, int RB, int NT, int S, bool UP2, bool ACT, int OS, bool DEC, bool HINT = false, int NWV = 4>
__device__ __forceinline__ void y5_conv_pw_body(const Y5ConvParams& p, const Y5HeadParams* hp) {
  typedef half_t T;
  static_assert(NT % OS == 0, "output split must divide the channel sub-tiles");
  constexpr int NPAD = 32 * NT;
  constexpr int NTH = NT / OS, NPH = 32 * NTH;  // channel sub-tiles / channels per epilogue group
  constexpr int NSLOT = RB / 16;          // 16-byte slots per LDS row of a K chunk
  constexpr int RPI = 1024 / RB;          // rows per LDS-DMA instruction
  constexpr int QN = 32 / RPI;            // instructions per chunk of a 32-row tile
  constexpr int STAGE_A = 32 * KC * RB, STAGE_O = 32 * NPH * 2;
  constexpr int STAGE = STAGE_A > STAGE_O ? STAGE_A : STAGE_O;
  constexpr int W_BYTES = KC * NPAD * RB;
  constexpr int LP = KC * QN;                               // loads per tile per wave
  constexpr int SPR = NPH / 8;                              // 16-byte slots per output row (of one epilogue group)
  constexpr int RPP = 64 / SPR;                             // output rows per store pass
  constexpr int NPASS = 32 / RPP;
  constexpr int SP = DEC ? 3 * ((32 * 85 / 8 + 63) / 64 + (HINT ? 1 : 0)) : NPASS * OS * (UP2 ? 5 : 1);  // stores per tile per wave
  constexpr int SWM = SPR >= 8 ? 7 : SPR - 1;               // scratch swizzle mask

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wlds = smem;
  float* blds = reinterpret_cast<float*>(smem + W_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = smem + W_BYTES + NPAD * 4 + wave * (S * STAGE);

  const T* __restrict__ xg = static_cast<const T*>(p.x);
  const T* __restrict__ wg = static_cast<const T*>(p.w);
  T* __restrict__ yg = static_cast<T*>(p.y);
  T* __restrict__ y2g = static_cast<T*>(p.y2);

  const int lrow = lane / NSLOT, lslot = lane % NSLOT;
  // ---- prologue: filter + bias into LDS (all waves), once per workgroup -------------------------------
  {
    constexpr int WI = NPAD * RB / 1024;  // LDS-DMA instructions per filter chunk
    for (int kc = 0; kc < KC; ++kc)
      for (int idx = wave; idx < WI; idx += NWV) {
        const int row = idx * RPI + lrow;
        const int sslot = lslot ^ Y5ConvGeom<T, RB>::swz(row);
        y5_glds16(reinterpret_cast<const char*>(wg + (size_t)row * p.Kpad) + kc * RB + sslot * 16, wlds + kc * NPAD * RB + idx * 1024);
      }
    for (int i = tid; i < NPAD; i += NWV * 64) blds[i] = p.bias[i];
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }

  // ---- tile schedule: workgroup tiles of 128 pixels, wave w takes rows [32w, 32w+32) --------------------
  const int G = gridDim.x, bid = blockIdx.x;
  const int nwt = p.M >> 5;                 // 32-pixel wave tiles (host guarantees M % 32 == 0)
  const int nbt = (nwt + NWV - 1) / NWV;
  const int nmine = (nbt - bid + G - 1) / G;
  auto tile_m0 = [&](int j) { return (y5_xcd_remap(bid + j * G, nbt) * NWV + wave) * 32; };
  int nw = nmine;                           // this wave's tile count (the last workgroup tile may be partial)
  if (nw > 0 && tile_m0(nw - 1) >= p.M) --nw;

  int aoff[QN];                             // per-lane byte offset of its 16-byte piece inside a tile, per instruction
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int row = q * RPI + lrow;
    aoff[q] = row * p.ldx * (int)sizeof(T) + (lslot ^ Y5ConvGeom<T, RB>::swz(row)) * 16;
  }
  auto issue = [&](int j, int buf) {
    const char* src = reinterpret_cast<const char*>(xg + (size_t)tile_m0(j) * p.ldx);
    char* dst = ring + buf * STAGE;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int q = 0; q < QN; ++q) y5_glds16(src + aoff[q] + kc * RB, dst + kc * 32 * RB + q * 1024);
  };

  const int g = lane >> 5, frow = lane & 31;
  const int fsw = Y5ConvGeom<T, RB>::swz(frow);
  const int orow = lane / SPR, oslot = lane % SPR;
  constexpr bool STATS = !ACT && !DEC && !UP2;   // the act = 0 instantiations serve the train-mode forward: optional BatchNorm statistics (p.bn_partial)
  Y5StatAcc stat[STATS ? OS : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int h = 0; h < OS; ++h) stat[h].clear();
  }

  for (int s = 0; s < S; ++s)
    if (s < nw) issue(s, s);

  int buf = 0;
  for (int i = 0; i < nw; ++i) {
    // ---- retire tile i's loads ----
    if (i + S - 1 >= nw) {
      y5_wait_vm<0>();
    } else if (i < S - 1) {
      switch (i) {
        case 0: y5_wait_vm<(S - 1) * LP>(); break;
        case 1: y5_wait_vm<(S - 1) * LP + SP>(); break;
        case 2: y5_wait_vm<(S - 1) * LP + 2 * SP>(); break;
        default: y5_wait_vm<(S - 1) * LP>(); break;
      }
    } else {
      y5_wait_vm<(S - 1) * (LP + SP)>();
    }
    __builtin_amdgcn_wave_barrier();  // (lock-step on hardware; orders the lanes of the host emulator)
    char* st = ring + buf * STAGE;
    // ---- MFMA ----
    // NT <= 2: two accumulators per output tile (alternating k-steps) so that MFMAs never wait on their own result
    constexpr bool DUAL = NT <= 2;
    float16_t acc[NT], acc2[DUAL ? NT : 1];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; if (DUAL) acc2[j][r] = 0.f; }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int ks = 0; ks < RB / 32; ++ks) {
        const int so = ((ks * 2 + g) ^ fsw) * 16;
        const half8_t af = *reinterpret_cast<const half8_t*>(st + kc * 32 * RB + frow * RB + so);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const half8_t wf = *reinterpret_cast<const half8_t*>(wlds + kc * NPAD * RB + (j * 32 + frow) * RB + so);
          if (DUAL && (ks & 1)) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc2[j], 0, 0, 0);
          else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc[j], 0, 0, 0);
        }
      }
    if constexpr (DUAL) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] += acc2[j];
    }
    // ---- epilogue: bias + act -> scratch (the vacated stage) -> full-row stores, one channel group at a time ----
    const int m0 = tile_m0(i);
    if constexpr (DEC) {
      // Fused Detect head (3 anchors x 85 outputs = channels 0..254 of the 256-channel tile).  Per anchor a: every lane drops the
      // values of its pixel (frow) whose channel lies in [85a, 85a + 85) into the scratch at [pixel][o] -- the fp16-rounded logit
      // for the four box outputs, sigmoid(logit) otherwise -- lanes 0..31 then turn the four box logits of their pixel into
      // xy / wh exactly as the decode kernel does, and the 32 x 85 halfs leave as 340 16-byte stores into the anchor's z rows.
      constexpr int NO = 85;
      static_assert(NT == 8 && STAGE >= 32 * NO * 2, "fused head: 256-channel tile, scratch of 32 x 85 halfs");
      const Y5HeadParams& hd = *hp;
      const int bimg = m0 / hd.npix;
      const int pix0 = m0 - bimg * hd.npix;
      half_t* sc = reinterpret_cast<half_t*>(st);
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
          const int pix = pix0 + lane;
          const int iy = (int)__umulhi((unsigned)pix, hd.inv_nx), ix = pix - iy * hd.nx;
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
        half_t* zrow = static_cast<half_t*>(hd.z) + ((long long)bimg * hd.nrows_total + hd.row_off + (long long)a * hd.npix + pix0) * NO;
        constexpr int NV = 32 * NO / 8;  // 16-byte vectors per anchor block
#pragma unroll
        for (int it = 0; it < (NV + 63) / 64; ++it) {
          const int v = it * 64 + lane;
          if (v < NV) *reinterpret_cast<uint4_t*>(zrow + v * 8) = *reinterpret_cast<const uint4_t*>(sc + v * 8);
          Y5_EMU_VM_OP(true);   // (it * 64 < NV: lane 0 always stores)
        }
        if constexpr (HINT) {  // the rows' objectness, bit for bit what z holds (one more store per anchor block: counted in SP)
          if (lane < 32)
            static_cast<half_t*>(hd.obj_hint)[(long long)bimg * hd.nrows_total + hd.row_off + (long long)a * hd.npix + pix0 + lane] = sc[lane * NO + 4];
          Y5_EMU_VM_OP(true);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the scratch is rewritten by the next anchor / refilled by the next tile's loads
      }
    } else
#pragma unroll
    for (int h = 0; h < OS; ++h) {
#pragma unroll
      for (int j = 0; j < NTH; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4_t bv = *reinterpret_cast<const float4_t*>(blds + (h * NTH + j) * 32 + q * 8 + g * 4);
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = acc[h * NTH + j][q * 4 + e] + bv[e];
            o[e] = (half_t)(ACT ? y5_silu(t) : t);
          }
          const int slot = j * 4 + q;
          *reinterpret_cast<half4_t*>(st + frow * (NPH * 2) + ((slot ^ (frow & SWM)) * 16) + g * 8) = o;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // lanes exchange data through the scratch: keep LDS writes before the reads
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int row = ps * RPP + orow;
        const uint4_t raw = *reinterpret_cast<const uint4_t*>(st + row * (NPH * 2) + ((oslot ^ (row & SWM)) * 16));
        const int m = m0 + row, n = h * NPH + oslot * 8;
        if constexpr (STATS) {
          if (p.bn_partial) stat[h].add(raw);   // (rows are always real pixels: M % 32 == 0; padded channels are dropped at the flush)
        }
        if (n < p.C2) {
          if (!UP2 && p.split_n) {
            T* d = n < p.split_n ? yg + (size_t)m * p.ldy + n : y2g + (size_t)m * p.ld2 + (n - p.split_n);
            *reinterpret_cast<uint4_t*>(d) = raw;
          } else
          *reinterpret_cast<uint4_t*>(yg + (size_t)m * p.ldy + n) = raw;
          if constexpr (UP2) {
            const int ohw = p.OH * p.OW;
            const int b = m / ohw;
            const int r = m - b * ohw;
            const int oh = r / p.OW, ow = r - oh * p.OW;
            const size_t row0 = ((size_t)b * 2 * p.OH + 2 * oh) * (2 * p.OW) + 2 * ow;
            T* d0 = y2g + row0 * p.ld2 + n;
            T* d1 = y2g + (row0 + 2 * p.OW) * p.ld2 + n;
            *reinterpret_cast<uint4_t*>(d0) = raw;
            *reinterpret_cast<uint4_t*>(d0 + p.ld2) = raw;
            *reinterpret_cast<uint4_t*>(d1) = raw;
            *reinterpret_cast<uint4_t*>(d1 + p.ld2) = raw;
          }
        }
#pragma unroll
        for (int st_i = 0; st_i < (UP2 ? 5 : 1); ++st_i) Y5_EMU_VM_OP(h * NPH < p.C2);   // (lane slot 0 holds the group's first channel)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the scratch is rewritten by the next group / refilled by the next tile's loads
    }
    // ---- refill the vacated stage with tile i+S ----
    if (i + S < nw) issue(i + S, buf);
    buf = buf + 1 == S ? 0 : buf + 1;
  }
  if constexpr (STATS) {
    if (p.bn_partial) {   // (kernel-uniform) every wave's ring is idle: its first stage carries the wave's sums to the cross-wave addition
      y5_wait_vm<0>();
      y5_stat_flush<SPR, OS, NWV>(stat, lane, tid, ring, smem + W_BYTES + NPAD * 4, S * STAGE, p.bn_partial + (size_t)blockIdx.x * 2 * p.C2, p.C2);
    }
  }
}

// NWV = 8: eight waves per workgroup under one LDS filter copy (two per SIMD), normally with a single stage each -- as conv_k3.h
template <int KC, int RB, int NT, int S, bool UP2, bool ACT = true, int OS = 1, int NWV = 4>
__global__ __launch_bounds__(NWV * 64)
void y5_conv_pw_kernel(const Y5ConvParams p) {
  y5_conv_pw_body<KC, RB, NT, S, UP2, ACT, OS, false, false, NWV>(p, nullptr);
}

// 1x1 Detect convolution (128 -> 3 x 85 channels) + Detect decode in one pass (export / z-only mode)
template <int KC, int RB, int NT, int S, int OS, bool HINT = false, int NWV = 4>
__global__ __launch_bounds__(NWV * 64)
void y5_conv_pw_head_kernel(const Y5ConvParams p, const Y5HeadParams h) {
  y5_conv_pw_body<KC, RB, NT, S, false, false, OS, true, HINT, NWV>(p, &h);
}
