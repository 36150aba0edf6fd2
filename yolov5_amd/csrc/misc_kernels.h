// HBM-bound glue kernels of the YOLOv5 forward path (gfx950): layout conversion at the API boundary,
// SPPF max-pool chain, nearest 2x upsample / channel-slice copies, Detect decode.
// Every kernel moves 16 bytes per lane where the layout allows it and is launched with >> 256 workgroups.
#pragma once
#include "y5_common.h"

// ---- element load/store helpers ---------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float y5_ld(const T* p) { return (float)*p; }
template <> __device__ __forceinline__ float y5_ld<unsigned char>(const unsigned char* p) { return (float)*p; }

// ---------------------------------------------------------------------------------------------------
// NCHW (u8|f16|f32) -> NHWC (f16|f32), channels padded to ld with zeros, values scaled (train.py:379)
// one thread per pixel: reads are coalesced per channel plane, the write is one ld-element vector
// ---------------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void y5_nchw_to_nhwc_kernel(const S* __restrict__ src, D* __restrict__ dst, int C, long long HW,
                                       long long npix, int ld, float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const long long b = i / HW, r = i - b * HW;
  const S* s = src + b * C * HW + r;
  D* d = dst + i * ld;
  for (int c = 0; c < ld; ++c) d[c] = c < C ? (D)(y5_ld<S>(s + (long long)c * HW) * scale) : (D)0.f;
}

// NHWC slice -> contiguous NCHW (API boundary only; one thread per output element, writes coalesced)
template <typename T>
__global__ void y5_nhwc_to_nchw_kernel(const T* __restrict__ src, T* __restrict__ dst, int C, long long HW,
                                       long long total, int ld) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long r = i % HW;
  const long long bc = i / HW;
  const long long c = bc % C, b = bc / C;
  dst[i] = src[(b * HW + r) * ld + c];
}

// ---------------------------------------------------------------------------------------------------
// SPPF pooling chain (models/common.py:338-340).  One workgroup per (image, 16-byte channel group):
// the H*W*16 B plane lives in LDS; three successive k x k stride-1 max pools (implicit -inf padding)
// are written to channel slices 1..3 of the same NHWC buffer.
// ---------------------------------------------------------------------------------------------------
// GV = 16-byte vectors of one pixel a workgroup owns (1 = a 16-byte channel group; 2 / 4 / 8 = 32 / 64 / 128 contiguous bytes per
// pixel instead of 16 bytes out of every 2 KiB-strided row -- against fewer, larger workgroups).
template <typename V, int GV, bool SEP = true>  // V = half8_t (8 channels) or float4_t (4 channels): 16 bytes; SEP: row pass + column pass (third LDS plane)
__global__ void y5_sppf_pool_kernel(char* __restrict__ buf, int H, int W, int C_bytes, int ld_bytes, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W;
  V* p0 = reinterpret_cast<V*>(smem);
  V* p1 = p0 + HW * GV;
  V* tmp = p1 + HW * GV;
  const int groups = C_bytes / (16 * GV);
  const int b = blockIdx.x / groups, cg = blockIdx.x - b * groups;
  char* base = buf + (size_t)b * HW * ld_bytes + (size_t)cg * 16 * GV;
  const int n = HW * GV;  // vector index v = pixel * GV + lane-in-pixel
  // every thread owns the SAME vector slots (v = tid + it * blockDim) in all six sub-passes: their (row, column) and clamped window bounds are
  // computed once (the first version re-derived them with two integer divisions per vector per sub-pass -- more instructions than the max chain itself)
  constexpr int MAXIT = 8;
  const int nit = (n + (int)blockDim.x - 1) / (int)blockDim.x;
  const int r = k / 2;
  if (nit > MAXIT || !SEP) {  // planes larger than MAXIT * blockDim vectors, or too large for the third LDS plane of the separable form (not a
                             // YOLOv5 P5 shape): generic index arithmetic, the k x k window directly
    for (int v = threadIdx.x; v < n; v += blockDim.x) p0[v] = *reinterpret_cast<const V*>(base + (size_t)(v / GV) * ld_bytes + (v % GV) * 16);
    __syncthreads();
    V* in = p0;
    V* out = p1;
    for (int pass = 1; pass <= 3; ++pass) {
      for (int v = threadIdx.x; v < n; v += blockDim.x) {
        const int i = v / GV, gl = v % GV;
        const int y = i / W, x = i - y * W;
        const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
        const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
        V m = in[(y0 * W + x0) * GV + gl];
        for (int yy = y0; yy <= y1; ++yy)
          for (int xx = x0; xx <= x1; ++xx) m = __builtin_elementwise_max(m, in[(yy * W + xx) * GV + gl]);
        out[v] = m;
        *reinterpret_cast<V*>(base + (size_t)i * ld_bytes + (size_t)pass * C_bytes + gl * 16) = m;
      }
      __syncthreads();
      V* t = in; in = out; out = t;
    }
    return;
  }
  int rowlo[MAXIT], rowcnt[MAXIT], collo[MAXIT], colcnt[MAXIT], pix[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int v = threadIdx.x + it * blockDim.x;
    const int vv = v < n ? v : 0;
    const int i = vv / GV, gl = vv % GV;
    const int y = i / W, x = i - y * W;
    const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
    const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
    rowlo[it] = (y * W + x0) * GV + gl; rowcnt[it] = v < n && it < nit ? x1 - x0 : -1;   // window = first element + cnt more, stride GV
    collo[it] = (y0 * W + x) * GV + gl; colcnt[it] = y1 - y0;                            // ... stride W * GV
    pix[it] = i;
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it)
    if (rowcnt[it] >= 0) {
      const int v = threadIdx.x + it * blockDim.x;
      p0[v] = *reinterpret_cast<const V*>(base + (size_t)pix[it] * ld_bytes + (v % GV) * 16);
    }
  __syncthreads();
  V* in = p0;
  V* out = p1;
  const int cs = W * GV;
  for (int pass = 1; pass <= 3; ++pass) {
    // max is exact, so the k x k window is taken separably: k reads along the row into `tmp`, k reads down the column (2k instead of k^2
    // LDS reads per output)
    {
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (rowcnt[it] >= 0) {
          const V* q = in + rowlo[it];
          V m = q[0];
          for (int j = 1; j <= rowcnt[it]; ++j) m = __builtin_elementwise_max(m, q[j * GV]);
          tmp[threadIdx.x + it * blockDim.x] = m;
        }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (rowcnt[it] >= 0) {
          const int v = threadIdx.x + it * blockDim.x;
          const V* q = tmp + collo[it];
          V m = q[0];
          for (int j = 1; j <= colcnt[it]; ++j) m = __builtin_elementwise_max(m, q[j * cs]);
          out[v] = m;
          *reinterpret_cast<V*>(base + (size_t)pix[it] * ld_bytes + (size_t)pass * C_bytes + (v % GV) * 16) = m;
        }
    }
    __syncthreads();
    V* t = in; in = out; out = t;
  }
}

// ---------------------------------------------------------------------------------------------------
// nearest 2x upsample into a channel slice; 16 B per lane.  dst pixel (b, y, x) <- src (b, y/2, x/2)
// ---------------------------------------------------------------------------------------------------
__global__ void y5_upsample2x_kernel(const char* __restrict__ src, char* __restrict__ dst, int H, int W, int vec_per_pix,
                                     int lds_bytes, int ldd_bytes, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over dst vectors
  if (i >= total) return;
  const int v = (int)(i % vec_per_pix);
  const long long pix = i / vec_per_pix;
  const int W2 = 2 * W, H2 = 2 * H;
  const int x = (int)(pix % W2);
  const long long t = pix / W2;
  const int y = (int)(t % H2);
  const long long b = t / H2;
  const long long sp = (b * H + (y >> 1)) * W + (x >> 1);
  *reinterpret_cast<uint4_t*>(dst + pix * ldd_bytes + v * 16) =
      *reinterpret_cast<const uint4_t*>(src + sp * lds_bytes + v * 16);
}

__global__ void y5_copy_slice_kernel(const char* __restrict__ src, char* __restrict__ dst, int vec_per_pix, int lds_bytes,
                                     int ldd_bytes, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int v = (int)(i % vec_per_pix);
  const long long pix = i / vec_per_pix;
  *reinterpret_cast<uint4_t*>(dst + pix * ldd_bytes + v * 16) =
      *reinterpret_cast<const uint4_t*>(src + pix * lds_bytes + v * 16);
}

// ---------------------------------------------------------------------------------------------------
// Detect decode (models/yolo.py:96-115).  One workgroup per (image, tile of P pixels): the tile's logits
// (P x ld, 16 B per lane, coalesced) are staged in LDS; then for every anchor the P*no output values, which are
// CONTIGUOUS in z (rows a*ny*nx + pix .. of image b) and in the raw (bs,na,ny,nx,no) tensor, are produced two
// per lane and written with 4-byte/8-byte stores.  Row index inside z is bit-exactly the reference's:
// row_off + a*ny*nx + iy*nx + ix.
// ---------------------------------------------------------------------------------------------------
struct Y5DecodeParams {
  const void* logits;
  void* z;
  void* raw;
  long long nrows_total, row_off;
  int ny, nx, na, no, nm, ld, P;
  unsigned inv_no;  // ceil(2^32 / no): idx / no == umulhi(idx, inv_no) for idx < 2^16
  unsigned inv_nx;  // ceil(2^32 / nx): pix / nx for pix < 2^16 (fast path only when ny*nx < 65536)
  float stride;
  float anchors_px[16];  // na*2
  void* obj_hint;        // optional (B, nrows_total) plane of z's dtype: a copy of every row's objectness for the NMS filter (y5_nms_batched_hint)
};

template <typename T>
__device__ __forceinline__ float y5_decode_one(const Y5DecodeParams& p, const T* tile, int pix0, int a, int e, float& rawv) {
  const int pl = (int)__umulhi((unsigned)e, p.inv_no);  // pixel inside the tile
  const int o = e - pl * p.no;
  const float v = (float)tile[pl * p.ld + a * p.no + o];
  rawv = v;
  if (o >= p.no - p.nm) return v;  // Segment mask coefficients are not activated (yolo.py:104-108)
  const float s = y5_sigmoid(v);
  if (o < 2) {
    const int pix = pix0 + pl;
    const int iy = pix / p.nx, ix = pix - iy * p.nx;
    const float g = (float)(o == 0 ? ix : iy) - 0.5f;  // grid = (ix-0.5, iy-0.5), yolo.py:126
    return (s * 2.0f + g) * p.stride;                   // yolo.py:110
  }
  if (o < 4) {
    const float w = s * 2.0f;
    return w * w * p.anchors_px[a * 2 + (o - 2)];  // yolo.py:111
  }
  return s;
}

template <typename T, typename Z>
__global__ __launch_bounds__(256)
void y5_detect_decode_kernel(const Y5DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);
  const int npix = p.ny * p.nx;
  const int b = blockIdx.y;
  const int pix0 = blockIdx.x * p.P;
  const int np = npix - pix0 < p.P ? npix - pix0 : p.P;
  // stage logits: np pixels x ld elements, contiguous in global memory
  {
    const uint4_t* src = reinterpret_cast<const uint4_t*>(static_cast<const T*>(p.logits) + ((long long)b * npix + pix0) * p.ld);
    uint4_t* dst = reinterpret_cast<uint4_t*>(smem);
    const int nvec = np * p.ld * (int)sizeof(T) / 16;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int ne = np * p.no;  // output elements per anchor for this tile
  // z-only fast path (the engine returns the raw tensors as views, export mode has none): the four box outputs of every
  // (pixel, anchor) row are computed FIRST and written back over their logits in the LDS tile (exact formulas of the general path,
  // already rounded to fp16); the element loop below then only needs a sigmoid per element and a pass-through select for o < 4
  // -- 12 instead of 38 VALU instructions per output.
  bool light = false;
  if constexpr (sizeof(T) == 2 && sizeof(Z) == 2) {
    light = p.z != nullptr && p.raw == nullptr && npix < 65536 && (ne & 7) == 0;
    for (int a = 0; a < p.na; ++a) {
      const long long zbase = ((long long)b * p.nrows_total + p.row_off + (long long)a * npix + pix0) * p.no;
      light = light && (zbase & 7) == 0;
    }
    if (light) {
      for (int rix = threadIdx.x; rix < np * p.na; rix += blockDim.x) {
        const int a = rix / np, pl = rix - a * np;
        const int pix = pix0 + pl;
        const int iy = (int)__umulhi((unsigned)pix, p.inv_nx), ix = pix - iy * p.nx;
        T* q = tile + pl * p.ld + a * p.no;
        const float gx = (float)ix - 0.5f, gy = (float)iy - 0.5f;
        const float aw = p.anchors_px[a * 2], ah = p.anchors_px[a * 2 + 1];
        float s2[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) s2[o] = __builtin_amdgcn_rcpf(1.0f + __expf(-(float)q[o])) * 2.0f;
        q[0] = (T)((s2[0] + gx) * p.stride);   // yolo.py:110
        q[1] = (T)((s2[1] + gy) * p.stride);
        q[2] = (T)(s2[2] * s2[2] * aw);         // yolo.py:111
        q[3] = (T)(s2[3] * s2[3] * ah);
      }
      __syncthreads();
    }
  }
  if (p.obj_hint && p.z) {
    // objectness of every (anchor, pixel) row with the formula of the path that writes z below (rcp form in the light path, true division
    // otherwise).  The NMS filter treats it as a HINT -- rows it cannot exclude with a margin are read from z -- so exactness is not required.
    for (int rix = threadIdx.x; rix < np * p.na; rix += blockDim.x) {
      const int a = rix / np, pl = rix - a * np;
      const float v = (float)tile[pl * p.ld + a * p.no + 4];
      const float sgm = light ? __builtin_amdgcn_rcpf(1.0f + __expf(-v)) : y5_sigmoid(v);
      static_cast<Z*>(p.obj_hint)[(long long)b * p.nrows_total + p.row_off + (long long)a * npix + pix0 + pl] = (Z)sgm;
    }
  }
  for (int a = 0; a < p.na; ++a) {
    const long long zbase = ((long long)b * p.nrows_total + p.row_off + (long long)a * npix + pix0) * p.no;
    const long long rbase = (((long long)b * p.na + a) * npix + pix0) * p.no;
    Z* zp = static_cast<Z*>(p.z) + zbase;
    T* rp = p.raw ? static_cast<T*>(p.raw) + rbase : nullptr;
    const bool pair_ok = ((zbase | rbase | ne) & 1) == 0;
    if constexpr (sizeof(T) == 2 && sizeof(Z) == 2) {
      if (light) {
        const int nact = p.no - p.nm;
        for (int e0 = threadIdx.x * 8; e0 < ne; e0 += blockDim.x * 8) {
          const int pl = (int)__umulhi((unsigned)e0, p.inv_no);
          int o = e0 - pl * p.no;
          const T* src = tile + pl * p.ld + a * p.no;
          half8_t zo;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float v = (float)src[o];
            const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
            zo[k] = (half_t)((o < 4 || o >= nact) ? v : s);  // box outputs are final already; Segment mask coefficients stay raw
            if (++o == p.no) { o = 0; src += p.ld; }
          }
          *reinterpret_cast<half8_t*>(zp + e0) = zo;
        }
        continue;
      }
      if (((zbase | rbase | ne) & 7) == 0 && npix < 65536) {
        // fast path: 8 consecutive outputs (16 bytes) per lane, branch-free decode.  a (hence the anchor) is uniform;
        // the 8 elements touch at most two pixels, whose grid coordinates are formed once with umulhi.
        const float aw = p.anchors_px[a * 2], ah = p.anchors_px[a * 2 + 1];
        const int nact = p.no - p.nm;
        for (int e0 = threadIdx.x * 8; e0 < ne; e0 += blockDim.x * 8) {
          int pl = (int)__umulhi((unsigned)e0, p.inv_no);
          int o = e0 - pl * p.no;
          const int pixA = pix0 + pl;
          const int iyA = (int)__umulhi((unsigned)pixA, p.inv_nx), ixA = pixA - iyA * p.nx;
          const int ixB = ixA + 1 == p.nx ? 0 : ixA + 1, iyB = ixA + 1 == p.nx ? iyA + 1 : iyA;
          float gx = (float)ixA - 0.5f, gy = (float)iyA - 0.5f;
          const float gxB = (float)ixB - 0.5f, gyB = (float)iyB - 0.5f;
          const T* src = tile + pl * p.ld + a * p.no;
          half8_t zo, ro;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float v = (float)src[o];
            const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
            const float s2 = s * 2.0f;
            const float rxy = (s2 + (o == 0 ? gx : gy)) * p.stride;   // yolo.py:110
            const float rwh = s2 * s2 * (o == 2 ? aw : ah);           // yolo.py:111
            float r = o < 2 ? rxy : (o < 4 ? rwh : s);
            r = o < nact ? r : v;                                     // Segment mask coefficients stay raw
            zo[k] = (half_t)r;
            ro[k] = (half_t)v;
            if (++o == p.no) { o = 0; src += p.ld; gx = gxB; gy = gyB; }
          }
          if (p.z) *reinterpret_cast<half8_t*>(zp + e0) = zo;
          if (rp) *reinterpret_cast<half8_t*>(rp + e0) = ro;
        }
        continue;
      }
    }
    if (pair_ok) {
      for (int e = threadIdx.x * 2; e < ne; e += blockDim.x * 2) {
        float r0, r1;
        const float v0 = y5_decode_one<T>(p, tile, pix0, a, e, r0);
        const float v1 = y5_decode_one<T>(p, tile, pix0, a, e + 1, r1);
        typedef Z Z2 __attribute__((ext_vector_type(2)));
        typedef T T2 __attribute__((ext_vector_type(2)));
        Z2 zo; zo[0] = (Z)v0; zo[1] = (Z)v1;
        if (p.z) *reinterpret_cast<Z2*>(zp + e) = zo;
        if (rp) { T2 ro; ro[0] = (T)r0; ro[1] = (T)r1; *reinterpret_cast<T2*>(rp + e) = ro; }
      }
    } else {
      for (int e = threadIdx.x; e < ne; e += blockDim.x) {
        float r0;
        const float v0 = y5_decode_one<T>(p, tile, pix0, a, e, r0);
        if (p.z) zp[e] = (Z)v0;
        if (rp) rp[e] = (T)r0;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Backward of the head layout change (models/yolo.py:96-98): draw (B, na, ny*nx, no) -> dlogits NHWC (B, ny*nx, ld),
// channels >= na*no zeroed.  One workgroup per (image, tile of P pixels): the na contiguous runs of P*no gradients are
// read 16 bytes per lane and scattered into an LDS tile laid out like the logits rows, which then leaves as 16-byte rows.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void y5_raw_to_nhwc_tiled_kernel(const half_t* __restrict__ draw, half_t* __restrict__ dlg, int npix, int na, int no, int ld, int P,
                                 unsigned inv_no) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* tile = reinterpret_cast<half_t*>(smem);
  const int b = blockIdx.y;
  const int pix0 = blockIdx.x * P;
  const int np = npix - pix0 < P ? npix - pix0 : P;
  for (int i = threadIdx.x; i < np * ld / 8; i += blockDim.x) reinterpret_cast<uint4_t*>(smem)[i] = uint4_t{0, 0, 0, 0};
  __syncthreads();
  const int ne = np * no;
  for (int a = 0; a < na; ++a) {
    const long long rbase = (((long long)b * na + a) * npix + pix0) * no;
    const half_t* src = draw + rbase;
    if (((rbase | ne) & 7) == 0) {
      for (int e0 = threadIdx.x * 8; e0 < ne; e0 += blockDim.x * 8) {
        const half8_t v = *reinterpret_cast<const half8_t*>(src + e0);
        int pl = (int)__umulhi((unsigned)e0, inv_no);
        int o = e0 - pl * no;
        half_t* dst = tile + pl * ld + a * no;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          dst[o] = v[k];
          if (++o == no) { o = 0; dst += ld; }
        }
      }
    } else {
      for (int e = threadIdx.x; e < ne; e += blockDim.x) {
        const int pl = (int)__umulhi((unsigned)e, inv_no);
        tile[pl * ld + a * no + (e - pl * no)] = src[e];
      }
    }
  }
  __syncthreads();
  uint4_t* out = reinterpret_cast<uint4_t*>(dlg + ((long long)b * npix + pix0) * ld);
  for (int i = threadIdx.x; i < np * ld / 8; i += blockDim.x) out[i] = reinterpret_cast<const uint4_t*>(smem)[i];
}
