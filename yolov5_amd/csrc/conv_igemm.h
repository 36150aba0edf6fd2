// NHWC implicit-GEMM convolution for gfx950 (MI355X): MFMA 32x32 tiles, LDS-DMA staged input/filter
// tiles (global_load_lds, 16 B/lane, XOR-swizzled on the SOURCE side), fused bias + SiLU + residual
// epilogue written straight from the accumulator registers (no LDS transpose).
//
// Replaces, for the reference hot path, `Conv.forward_fuse` (models/common.py:90-92: act(conv(x)) with the
// BN folded by utils/torch_utils.py:224-254), the residual add of `Bottleneck.forward` (common.py:181),
// the channel concat of `C3`/`SPPF`/`Concat` (common.py:246,340,453: the output is written into a channel
// slice [c_off, c_off+C2) of a wider NHWC buffer, pixel stride ldy) and `nn.Upsample(2,'nearest')`
// (yolov5s.yaml:36,41: optional second, 2x-replicated store).
//
// GEMM view (operands swapped so that one lane owns ONE output pixel and 4 consecutive channels):
//     D[n][m] = sum_k  Wp[n][k] * A[m][k]      n = output channel, m = output pixel (b,oh,ow),
//     k = (kh, kw, c) flattened, A gathered on the fly from x[b][oh*SH-PH+kh][ow*SW-PW+kw][c].
// MFMA "A" operand = filter rows, "B" operand = activation rows; both are 64-byte LDS rows (32 halfs /
// 16 floats of K) read with one ds_read_b128 per lane: lane l -> row (l & 31), 16-byte slot (l >> 5).
// Any permutation of k inside a 64-byte row is legal as long as filter and activation use the same one.
//
// LDS row swizzle: 16-byte slot s of row r is stored at slot s ^ ((r >> 2) & 3).  With 64-byte rows the
// four 16-lane groups of a ds_read_b128 then hit 16 distinct 16-byte bank slots (conflict free).  The LDS-DMA
// destination must stay lane-linear, so the permutation is applied to the per-lane GLOBAL source address.
#pragma once
#include "y5_common.h"

struct Y5ConvParams {
  const void* x;      // input  NHWC, pixel stride ldx elements
  const void* w;      // packed filter [Npad][Kpad], k = (kh,kw,c)
  const float* bias;  // [Npad] fp32
  const void* res;    // optional residual, same geometry as y (pixel stride ldr), may alias y
  void* y;            // output NHWC slice, pixel stride ldy
  void* y2;           // optional second destination: 2x nearest-upsampled copy (pixel stride ld2)
  const void* zero;   // >= 64 bytes of zeros in global memory (source for padding taps / tails)
  int B, H, W, C1, ldx;
  int OH, OW, C2, ldy;
  int KH, KW, SH, SW, PH, PW;
  int act;  // 0 = identity, 1 = SiLU
  int Kpad, Npad, K;
  int ldr, ld2;
  int M;  // B*OH*OW
  int tilesM, tilesN, nk;
};

template <typename T> struct Y5Tr;
template <> struct Y5Tr<half_t> { static constexpr int EPP = 8, BK = 32; };
template <> struct Y5Tr<float>  { static constexpr int EPP = 4, BK = 16; };

#define Y5_CONV_ROWB 64       // bytes per LDS row (= BK elements)
#define Y5_CONV_MAXTAB 2048   // max k-pieces in TABLE mode (LDS: 8 B each)

template <typename T, int WM, int WN, int TM, int TN, bool TABLE>
__global__ __launch_bounds__(WM * WN * 64)
void y5_conv_igemm_kernel(const Y5ConvParams p) {
  using Tr = Y5Tr<T>;
  constexpr int EPP = Tr::EPP, BK = Tr::BK;
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int ACT_INSTR = BM / 16, WGT_INSTR = BN / 16;
  static_assert(ACT_INSTR % NW == 0, "activation tile must split evenly over the waves");
  constexpr int ACT_PER_WAVE = ACT_INSTR / NW;
  constexpr int WGT_PER_WAVE = (WGT_INSTR + NW - 1) / NW;
  constexpr int BUF_BYTES = (BM + BN) * Y5_CONV_ROWB;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [buf0: act BM rows | wgt BN rows][buf1: same][tap table (TABLE mode only)]
  int2* tab = reinterpret_cast<int2*>(smem + 2 * BUF_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tile = y5_xcd_remap(blockIdx.x, p.tilesM * p.tilesN);
  const int tile_n = tile % p.tilesN, tile_m = tile / p.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ xg = static_cast<const T*>(p.x);
  const T* __restrict__ wg = static_cast<const T*>(p.w);
  const char* zero = static_cast<const char*>(p.zero);

  if constexpr (TABLE) {
    // per k-piece (EPP elements) gather table: {element offset relative to pixel (ih0,iw0), kh | kw<<16}
    const int npieces = p.Kpad / EPP;
    for (int q = tid; q < npieces; q += NW * 64) {
      const int k = q * EPP;
      int2 e;
      if (k < p.K) {
        const int tap = k / p.C1, c = k - tap * p.C1;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        e.x = (kh * p.W + kw) * p.ldx + c;
        e.y = kh | (kw << 16);
      } else {
        e.x = 0;
        e.y = 0x7fff;  // kh = 32767 -> never inside the image
      }
      tab[q] = e;
    }
  }

  // ---- per-lane loader state ---------------------------------------------------------------------
  const int lrow = lane >> 2;   // row inside one 16-row LDS-DMA instruction
  const int lslot = lane & 3;   // destination 16-byte slot
  int a_base[ACT_PER_WAVE];     // element offset of (b, ih0, iw0, source slot) ; meaningless if !a_ok
  int a_ih0[ACT_PER_WAVE], a_iw0[ACT_PER_WAVE];
  int a_slot[ACT_PER_WAVE];
#pragma unroll
  for (int i = 0; i < ACT_PER_WAVE; ++i) {
    const int row = (wave + i * NW) * 16 + lrow;  // row inside the activation tile
    const int sslot = lslot ^ ((row >> 2) & 3);
    const int m = m0 + row;
    const int mm = m < p.M ? m : 0;
    const int ohw = p.OH * p.OW;
    const int b = mm / ohw;
    const int r = mm - b * ohw;
    const int oh = r / p.OW, ow = r - oh * p.OW;
    const int ih0 = oh * p.SH - p.PH, iw0 = ow * p.SW - p.PW;
    a_ih0[i] = m < p.M ? ih0 : -0x40000000;  // pixel rows past M: every tap reads zeros
    a_iw0[i] = iw0;
    a_base[i] = ((b * p.H + ih0) * p.W + iw0) * p.ldx + sslot * EPP;
    a_slot[i] = sslot;
  }
  const char* w_src[WGT_PER_WAVE];
  bool w_ok[WGT_PER_WAVE];
#pragma unroll
  for (int i = 0; i < WGT_PER_WAVE; ++i) {
    const int row = (wave + i * NW) * 16 + lrow;  // row inside the filter tile
    const int sslot = lslot ^ ((row >> 2) & 3);
    const int n = n0 + row;
    w_ok[i] = (row < BN) && (n < p.Npad);
    w_src[i] = reinterpret_cast<const char*>(wg + (size_t)(w_ok[i] ? n : 0) * p.Kpad + sslot * EPP);
  }

  if constexpr (TABLE) __syncthreads();

  // uniform tap walker (UNIFORM mode: C1 % BK == 0 so one BK chunk never straddles a filter tap)
  int u_kh = 0, u_kw = 0, u_c0 = 0;

  auto stage = [&](int kc, int buf) {
    char* lds = smem + buf * BUF_BYTES;
    int tap_off = 0;
    if constexpr (!TABLE) tap_off = (u_kh * p.W + u_kw) * p.ldx + u_c0;
#pragma unroll
    for (int i = 0; i < ACT_PER_WAVE; ++i) {
      int off, ih, iw;
      if constexpr (TABLE) {
        const int2 e = tab[kc * 4 + a_slot[i]];
        off = a_base[i] - a_slot[i] * EPP + e.x;
        ih = a_ih0[i] + (e.y & 0xffff);
        iw = a_iw0[i] + (e.y >> 16);
      } else {
        off = a_base[i] + tap_off;
        ih = a_ih0[i] + u_kh;
        iw = a_iw0[i] + u_kw;
      }
      const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const char* src = ok ? reinterpret_cast<const char*>(xg + off) : zero;
      y5_glds16(src, lds + (wave + i * NW) * 1024);
    }
#pragma unroll
    for (int i = 0; i < WGT_PER_WAVE; ++i) {
      const int idx = wave + i * NW;
      if (idx < WGT_INSTR) {
        const char* src = w_ok[i] ? w_src[i] + (size_t)kc * BK * sizeof(T) : zero;
        y5_glds16(src, lds + BM * Y5_CONV_ROWB + idx * 1024);
      }
    }
    if constexpr (!TABLE) {  // advance the uniform tap walker by one BK chunk
      u_c0 += BK;
      if (u_c0 >= p.C1) {
        u_c0 = 0;
        if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
      }
    }
  };

  float16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (bytes inside a buffer); lane -> row (lane&31), k-slot group g = lane>>5
  const int g = lane >> 5;
  const int frow = lane & 31;
  int a_rd[TM], w_rd[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_rd[i] = (wm * TM * 32 + i * 32 + frow) * Y5_CONV_ROWB;
#pragma unroll
  for (int j = 0; j < TN; ++j) w_rd[j] = BM * Y5_CONV_ROWB + (wn * TN * 32 + j * 32 + frow) * Y5_CONV_ROWB;
  const int fsw = (frow >> 2) & 3;  // swizzle term of this lane's rows (tile bases are multiples of 32)

  stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): LDS-DMA of the next tile has landed
  __syncthreads();

  const int nk = p.nk;
  for (int kc = 0; kc < nk; ++kc) {
    const int cur = kc & 1;
    if (kc + 1 < nk) stage(kc + 1, cur ^ 1);
    const char* lds = smem + cur * BUF_BYTES;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = ((ks * 2 + g) ^ fsw) * 16;
        half8_t af[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8_t*>(lds + a_rd[i] + so);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const half8_t*>(lds + w_rd[j] + so);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    } else {
      const int so0 = ((2 * g) ^ fsw) * 16, so1 = ((2 * g + 1) ^ fsw) * 16;
      float4_t af[TM][2], wf[TN][2];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        af[i][0] = *reinterpret_cast<const float4_t*>(lds + a_rd[i] + so0);
        af[i][1] = *reinterpret_cast<const float4_t*>(lds + a_rd[i] + so1);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        wf[j][0] = *reinterpret_cast<const float4_t*>(lds + w_rd[j] + so0);
        wf[j][1] = *reinterpret_cast<const float4_t*>(lds + w_rd[j] + so1);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][h][e], af[i][h][e], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): LDS-DMA of the next tile has landed
    __syncthreads();
  }

  // ---- epilogue: D[n][m]; lane owns pixel m = .. + (lane & 31) and channels 8*q + 4*(lane>>5) + {0..3}
  T* yg = static_cast<T*>(p.y);  // may alias p.res (in-place residual): no restrict
  const T* rg = static_cast<const T*>(p.res);
  T* y2g = static_cast<T*>(p.y2);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * TM * 32 + i * 32 + frow;
    if (m >= p.M) continue;
    size_t up_off[4];
    if (y2g) {
      const int ohw = p.OH * p.OW;
      const int b = m / ohw;
      const int r = m - b * ohw;
      const int oh = r / p.OW, ow = r - oh * p.OW;
      const size_t row0 = ((size_t)b * 2 * p.OH + 2 * oh) * (2 * p.OW) + 2 * ow;
      up_off[0] = row0 * p.ld2;
      up_off[1] = (row0 + 1) * p.ld2;
      up_off[2] = (row0 + 2 * p.OW) * p.ld2;
      up_off[3] = (row0 + 2 * p.OW + 1) * p.ld2;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * TN * 32 + j * 32 + q * 8 + g * 4;
        if (n >= p.C2) continue;
        const float4_t bv = *reinterpret_cast<const float4_t*>(p.bias + n);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[i][j][q * 4 + e] + bv[e];
          v[e] = p.act ? y5_silu(t) : t;
        }
        if (rg) {
          const T* rp = rg + (size_t)m * p.ldr + n;
          if constexpr (sizeof(T) == 2) {
            const half4_t rv = *reinterpret_cast<const half4_t*>(rp);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
          } else {
            const float4_t rv = *reinterpret_cast<const float4_t*>(rp);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
          }
        }
        if constexpr (sizeof(T) == 2) {
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
          *reinterpret_cast<half4_t*>(yg + (size_t)m * p.ldy + n) = o;
          if (y2g) {
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<half4_t*>(y2g + up_off[u] + n) = o;
          }
        } else {
          float4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = v[e];
          *reinterpret_cast<float4_t*>(yg + (size_t)m * p.ldy + n) = o;
          if (y2g) {
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<float4_t*>(y2g + up_off[u] + n) = o;
          }
        }
      }
    }
  }
}
