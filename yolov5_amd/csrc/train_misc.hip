// Small HBM-bound kernels of the training path (fp16 NHWC slices): head layout changes between the NHWC logits of the
// Detect convolutions and the (bs, na, ny, nx, no) tensors ComputeLoss consumes (models/yolo.py:96-98 view+permute),
// and the backward of nn.Upsample(2,'nearest'), of the Bottleneck shortcut / Concat fan-out (gradient accumulation)
// and of SPPF's three chained MaxPool2d(k,1,k//2) (models/common.py:338-340).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "../../include/yolov5_hip.h"
#include "y5_common.h"
#include "y5_host.h"

namespace {
inline unsigned nblk(long long n, int per) { long long b = (n + per - 1) / per; return (unsigned)(b < 1 ? 1 : b); }
}

// logits (B, npix, ld) -> raw (B, na, npix, no): raw[b][a][pix][o] = logits[b][pix][a*no + o]
__global__ void y5_nhwc_to_raw_kernel(const half_t* __restrict__ lg, half_t* __restrict__ raw, int npix, int na, int no, int ld, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over raw elements
  if (i >= total) return;
  const int o = (int)(i % no);
  long long t = i / no;
  const int pix = (int)(t % npix);
  t /= npix;
  const int a = (int)(t % na);
  const long long b = t / na;
  raw[i] = lg[(b * npix + pix) * ld + a * no + o];
}
// draw (B, na, npix, no) -> dlogits (B, npix, ld), channels >= na*no zero
__global__ void y5_raw_to_nhwc_kernel(const half_t* __restrict__ draw, half_t* __restrict__ dlg, int npix, int na, int no, int ld, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over dlogits elements
  if (i >= total) return;
  const int n = (int)(i % ld);
  const long long bp = i / ld;
  half_t v = (half_t)0.f;
  if (n < na * no) {
    const int a = n / no, o = n - a * no;
    const long long b = bp / npix;
    const int pix = (int)(bp - b * npix);
    v = draw[((b * na + a) * npix + pix) * no + o];
  }
  dlg[i] = v;
}

// dst(b,h,w,:) (+)= sum of the 2x2 block of src(b,2h..2h+1,2w..2w+1,:)
__global__ void y5_upsample2x_bwd_kernel(const char* __restrict__ src, char* __restrict__ dst, int H, int W, int vpp, int lds_b, int ldd_b,
                                         int acc, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over dst vectors
  if (i >= total) return;
  const int v = (int)(i % vpp);
  const long long pix = i / vpp;
  const int w = (int)(pix % W);
  const long long t = pix / W;
  const int h = (int)(t % H);
  const long long b = t / H;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (acc) {
    const half8_t d = *reinterpret_cast<const half8_t*>(dst + pix * ldd_b + v * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = (float)d[e];
  }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const long long sp = (b * 2 * H + 2 * h + dy) * (2 * W) + 2 * w + dx;
      const half8_t q = *reinterpret_cast<const half8_t*>(src + sp * lds_b + v * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)q[e];
    }
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)s[e];
  *reinterpret_cast<half8_t*>(dst + pix * ldd_b + v * 16) = o;
}

__global__ void y5_add_slice_kernel(const char* __restrict__ src, char* __restrict__ dst, int vpp, int lds_b, int ldd_b, int acc, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int v = (int)(i % vpp);
  const long long pix = i / vpp;
  half8_t q = *reinterpret_cast<const half8_t*>(src + pix * lds_b + v * 16);
  if (acc) {
    const half8_t d = *reinterpret_cast<const half8_t*>(dst + pix * ldd_b + v * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = (half_t)((float)q[e] + (float)d[e]);
  }
  *reinterpret_cast<half8_t*>(dst + pix * ldd_b + v * 16) = q;
}

// SPPF backward.  act: NHWC buffer [x | y1 | y2 | y3] (4*C channels), grad: same geometry holding d/d[x|y1|y2|y3] as left by
// the consumer's data-gradient; on return grad[..., 0:C] holds the total gradient w.r.t. x (slices 1..3 are consumed).
// One workgroup per (image, 8-channel group); planes in LDS; argmax = first maximum in (kh, kw) scan order (torch's max_pool2d_with_indices tie rule).
// DETERMINISTIC, two forms (the first version scattered with fp32 LDS atomics, whose order -- hence rounding -- changed from run to run:
// profiles/r05/r05_sppf_bwd_deterministic.log):
//   y5_sppf_pool_bwd_kernel        every pool output scatters its gradient to the position of its maximum with 64-bit INTEGER LDS atomics on a 2^-24
//                                  fixed-point grid.  Every fp16 value is a multiple of 2^-24, so are the sums: the accumulation is exact, hence independent
//                                  of the order of the adds; between passes the total is rounded to fp32 once.  112 bytes of LDS per 8-channel vector.
//   y5_sppf_pool_bwd_gather_kernel fallback for planes that do not fit (88 bytes per vector): every output records the window-relative position of its
//                                  maximum (one byte), every INPUT position sums, in scan order, the outputs whose maximum it is.  2.4x slower.
// GV = 16-byte channel groups (8 channels each) a workgroup owns.
__device__ __forceinline__ long long y5_fix24(float x) { return (long long)(x * 16777216.0f); }   // exact: x is a multiple of 2^-24, |x| < 2^38
__device__ __forceinline__ float y5_unfix24(long long s) { return (float)s * (1.0f / 16777216.0f); }   // one rounding (to nearest) of the exact sum

template <int GV>
__global__ __launch_bounds__(256)
void y5_sppf_pool_bwd_kernel(const char* __restrict__ act, char* __restrict__ grad, int H, int W, int C_bytes, int lda_b, int ldg_b, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W;
  const int n = HW * GV;                                               // vector index v = pixel * GV + group-in-workgroup
  half8_t* a_in = reinterpret_cast<half8_t*>(smem);                 // [n] activation of the pool input
  // component-major planes [8][n]: consecutive lanes hit consecutive LDS banks
  float* g_out = reinterpret_cast<float*>(a_in + n);                // [8][n] gradient of the pool output (fp32: a multiple of 2^-24)
  unsigned long long* g_in = reinterpret_cast<unsigned long long*>(g_out + (size_t)n * 8);   // [8][n] gradient of the pool input, 2^-24 fixed point
  // Inf / NaN cannot live on the integer grid: a non-finite incoming gradient (an fp16 overflow that the loss scaler must SEE in order to skip the step)
  // poisons the workgroup's whole output with NaN instead of turning into a large finite number
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  int bad = 0;
  const int groups = C_bytes / (16 * GV);
  const int b = blockIdx.x / groups, cg = blockIdx.x - b * groups;
  const char* abase = act + (size_t)b * HW * lda_b + (size_t)cg * 16 * GV;
  char* gbase = grad + (size_t)b * HW * ldg_b + (size_t)cg * 16 * GV;
  const int r = k / 2;
  // g_out <- d/dy3
  for (int v = threadIdx.x; v < n; v += blockDim.x) {
    const half8_t q = *reinterpret_cast<const half8_t*>(gbase + (size_t)(v / GV) * ldg_b + 3 * (size_t)C_bytes + (v % GV) * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) { float f = (float)q[e]; if (!(fabsf(f) <= 65504.0f)) { bad = 1; f = 0.f; } g_out[e * n + v] = f; }
  }
  for (int pass = 3; pass >= 1; --pass) {  // pool `pass`: input slice pass-1 -> output slice pass
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
      const size_t po = (size_t)(pass - 1) * C_bytes + (v % GV) * 16;
      a_in[v] = *reinterpret_cast<const half8_t*>(abase + (size_t)(v / GV) * lda_b + po);
      const half8_t q = *reinterpret_cast<const half8_t*>(gbase + (size_t)(v / GV) * ldg_b + po);
#pragma unroll
      for (int e = 0; e < 8; ++e) {   // direct gradient of that slice (from cv2's data-gradient)
        float f = (float)q[e];
        if (!(fabsf(f) <= 65504.0f)) { bad = 1; f = 0.f; }
        g_in[e * n + v] = (unsigned long long)y5_fix24(f);
      }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
      const int i = v / GV, gl = v % GV;
      const int y = i / W, x = i - y * W;
      const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
      const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
      // one LDS read per window position, eight running (maximum, position) pairs; strict > keeps the FIRST maximum in scan order
      float best[8];
      int bi[8];
      {
        const half8_t q = a_in[(y0 * W + x0) * GV + gl];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = (float)q[e]; bi[e] = y0 * W + x0; }
      }
      for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) {
          const int pos = yy * W + xx;
          const half8_t q = a_in[pos * GV + gl];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = (float)q[e];
            if (t > best[e]) { best[e] = t; bi[e] = pos; }
          }
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(&g_in[e * n + bi[e] * GV + gl], (unsigned long long)y5_fix24(g_out[e * n + v]));   // integer: any order, same sum
    }
    __syncthreads();
    // the accumulated input gradient is the next pass's output gradient
    for (int q = threadIdx.x; q < n * 8; q += blockDim.x) g_out[q] = y5_unfix24((long long)g_in[q]);
    __syncthreads();
  }
  if (bad) s_bad = 1;   // (benign race: every writer stores the same value)
  __syncthreads();
  const bool poison = s_bad != 0;
  for (int v = threadIdx.x; v < n; v += blockDim.x) {
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = poison ? (half_t)__builtin_nanf("") : (half_t)g_out[e * n + v];
    *reinterpret_cast<half8_t*>(gbase + (size_t)(v / GV) * ldg_b + (v % GV) * 16) = o;
  }
}

template <int GV>
__global__ __launch_bounds__(256)
void y5_sppf_pool_bwd_gather_kernel(const char* __restrict__ act, char* __restrict__ grad, int H, int W, int C_bytes, int lda_b, int ldg_b, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W;
  const int n = HW * GV;                                               // vector index v = pixel * GV + group-in-workgroup
  half8_t* a_in = reinterpret_cast<half8_t*>(smem);                 // [n] activation of the pool input
  // component-major planes [8][n]: consecutive lanes hit consecutive LDS banks (the [n][8] layout put a wave's accesses 32 bytes apart: a
  // 16-way bank conflict on every one of the eight accesses per vector -- 246 us for a 26 MB tensor)
  float* g_out = reinterpret_cast<float*>(a_in + n);                // [8][n] gradient of the pool output
  float* g_in = g_out + (size_t)n * 8;                              // [8][n] gradient of the pool input (direct part, then the total)
  unsigned char* amax = reinterpret_cast<unsigned char*>(g_in + (size_t)n * 8);   // [8][n] window-relative position (dy * k + dx, k <= 15) of the maximum of output v's window
  const int groups = C_bytes / (16 * GV);
  const int b = blockIdx.x / groups, cg = blockIdx.x - b * groups;
  const char* abase = act + (size_t)b * HW * lda_b + (size_t)cg * 16 * GV;
  char* gbase = grad + (size_t)b * HW * ldg_b + (size_t)cg * 16 * GV;
  const int r = k / 2;
  // g_out <- d/dy3
  for (int v = threadIdx.x; v < n; v += blockDim.x) {
    const half8_t q = *reinterpret_cast<const half8_t*>(gbase + (size_t)(v / GV) * ldg_b + 3 * (size_t)C_bytes + (v % GV) * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) g_out[e * n + v] = (float)q[e];
  }
  for (int pass = 3; pass >= 1; --pass) {  // pool `pass`: input slice pass-1 -> output slice pass
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
      const size_t po = (size_t)(pass - 1) * C_bytes + (v % GV) * 16;
      a_in[v] = *reinterpret_cast<const half8_t*>(abase + (size_t)(v / GV) * lda_b + po);
      const half8_t q = *reinterpret_cast<const half8_t*>(gbase + (size_t)(v / GV) * ldg_b + po);
#pragma unroll
      for (int e = 0; e < 8; ++e) g_in[e * n + v] = (float)q[e];  // direct gradient of that slice (from cv2's data-gradient)
    }
    __syncthreads();
    // (a) where is the maximum of output v's window
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
      const int i = v / GV, gl = v % GV;
      const int y = i / W, x = i - y * W;
      const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
      const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
      // one LDS read per window position, eight running (maximum, position) pairs; strict > keeps the FIRST maximum in scan order
      float best[8];
      int bi[8];
      {
        const half8_t q = a_in[(y0 * W + x0) * GV + gl];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = (float)q[e]; bi[e] = (y0 - y + r) * k + (x0 - x + r); }
      }
      for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) {
          const int pos = yy * W + xx, code = (yy - y + r) * k + (xx - x + r);   // window-relative position: what phase (b) compares with
          const half8_t q = a_in[pos * GV + gl];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = (float)q[e];
            if (t > best[e]) { best[e] = t; bi[e] = code; }
          }
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) amax[e * n + v] = (unsigned char)bi[e];
    }
    __syncthreads();
    // (b) input position i collects, in scan order, the gradients of the outputs whose window contains it and whose maximum it is
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
      const int i = v / GV, gl = v % GV;
      const int y = i / W, x = i - y * W;
      const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
      const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
      float s[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = g_in[e * n + v];
      for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) {
          const int ov = (yy * W + xx) * GV + gl;
          const unsigned char me = (unsigned char)((y - yy + r) * k + (x - xx + r));   // this input position as seen from output (yy, xx)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (amax[e * n + ov] == me) s[e] += g_out[e * n + ov];
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) g_in[e * n + v] = s[e];   // (nobody else reads or writes this slot in this phase)
    }
    __syncthreads();
    // the accumulated input gradient is the next pass's output gradient
    for (int q = threadIdx.x; q < n * 8; q += blockDim.x) g_out[q] = g_in[q];
    __syncthreads();
  }
  for (int v = threadIdx.x; v < n; v += blockDim.x) {
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)g_out[e * n + v];
    *reinterpret_cast<half8_t*>(gbase + (size_t)(v / GV) * ldg_b + (v % GV) * 16) = o;
  }
}

// ---- fp32 twins (TrainEngine's reference-precision mode: the whole step in fp32 for an exact gradient check against the oracle's
// autograd; simple one-element-per-thread kernels, nothing here is on a hot path) ------------------------------------------------
__global__ void y5_nhwc_to_raw_f32_kernel(const float* __restrict__ lg, float* __restrict__ raw, int npix, int na, int no, int ld, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int o = (int)(i % no);
  long long t = i / no;
  const int pix = (int)(t % npix);
  t /= npix;
  const int a = (int)(t % na);
  const long long b = t / na;
  raw[i] = lg[(b * npix + pix) * ld + a * no + o];
}
__global__ void y5_raw_to_nhwc_f32_kernel(const float* __restrict__ draw, float* __restrict__ dlg, int npix, int na, int no, int ld, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i % ld);
  const long long bp = i / ld;
  float v = 0.f;
  if (n < na * no) {
    const int a = n / no, o = n - a * no;
    const long long b = bp / npix;
    const int pix = (int)(bp - b * npix);
    v = draw[((b * na + a) * npix + pix) * no + o];
  }
  dlg[i] = v;
}
__global__ void y5_upsample2x_bwd_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C, int lds, int ldd, int acc,
                                             long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over dst elements (pixel, channel)
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pix = i / C;
  const int w = (int)(pix % W);
  const long long t = pix / W;
  const int h = (int)(t % H);
  const long long b = t / H;
  float s = acc ? dst[pix * ldd + c] : 0.f;
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) s += src[((b * 2 * H + 2 * h + dy) * (2 * W) + 2 * w + dx) * lds + c];
  dst[pix * ldd + c] = s;
}
__global__ void y5_add_slice_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int lds, int ldd, int acc, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pix = i / C;
  dst[pix * ldd + c] = src[pix * lds + c] + (acc ? dst[pix * ldd + c] : 0.f);
}
// SPPF backward in place, one thread per (image, channel): the gradient of pool `pass` (held in slice `pass`) is scattered to the first
// maximum of every window of slice pass - 1 (torch's tie rule) on top of that slice's direct gradient; after pass 1 slice 0 holds dx.
__global__ void y5_sppf_pool_bwd_f32_kernel(const float* __restrict__ act, float* __restrict__ grad, int H, int W, int C, int lda, int ldg, int k, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C, b = i / C, r = k / 2;
  const float* a0 = act + (size_t)b * H * W * lda + c;
  float* g0 = grad + (size_t)b * H * W * ldg + c;
  for (int pass = 3; pass >= 1; --pass) {
    const float* ain = a0 + (size_t)(pass - 1) * C;
    float* gin = g0 + (size_t)(pass - 1) * C;
    const float* gout = g0 + (size_t)pass * C;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r, x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
        float best = ain[(size_t)(y0 * W + x0) * lda];
        int bi = y0 * W + x0;
        for (int yy = y0; yy <= y1; ++yy)
          for (int xx = x0; xx <= x1; ++xx) {
            const float t = ain[(size_t)(yy * W + xx) * lda];
            if (t > best) { best = t; bi = yy * W + xx; }
          }
        gin[(size_t)bi * ldg] += gout[(size_t)(y * W + x) * ldg];
      }
  }
}

extern "C" int y5_train_glue_f32(int op, const void* src, void* dst, int B, int H, int W, int C, int a, int b2, int c3, void* stream_) {
  // op 0: nhwc_to_raw (H = npix, W = na, C = no, a = ld)      op 1: raw_to_nhwc (same)
  // op 2: upsample2x_bwd (a = ld_up, b2 = ld_src, c3 = accumulate)   op 3: add_slice (B*H*W pixels; a = lds, b2 = ldd, c3 = accumulate)
  // op 4: sppf_pool_bwd (src = act, dst = grad; a = ld_act, b2 = ld_grad, c3 = k)
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!src || !dst) return y5_fail(Y5_ERR_BAD_ARG, "train_glue_f32: null pointer");
  if (op == 0) {
    const long long total = (long long)B * W * H * C;
    hipLaunchKernelGGL(y5_nhwc_to_raw_f32_kernel, dim3(nblk(total, 256)), dim3(256), 0, st, (const float*)src, (float*)dst, H, W, C, a, total);
  } else if (op == 1) {
    const long long total = (long long)B * H * a;
    hipLaunchKernelGGL(y5_raw_to_nhwc_f32_kernel, dim3(nblk(total, 256)), dim3(256), 0, st, (const float*)src, (float*)dst, H, W, C, a, total);
  } else if (op == 2) {
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(y5_upsample2x_bwd_f32_kernel, dim3(nblk(total, 256)), dim3(256), 0, st, (const float*)src, (float*)dst, H, W, C, a, b2, c3, total);
  } else if (op == 3) {
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(y5_add_slice_f32_kernel, dim3(nblk(total, 256)), dim3(256), 0, st, (const float*)src, (float*)dst, C, a, b2, c3, total);
  } else if (op == 4) {
    const int total = B * C;
    hipLaunchKernelGGL(y5_sppf_pool_bwd_f32_kernel, dim3(nblk(total, 64)), dim3(64), 0, st, (const float*)src, (float*)dst, H, W, C, a, b2, c3, total);
  } else {
    return y5_fail(Y5_ERR_BAD_ARG, "train_glue_f32: unknown op");
  }
  return y5_check_launch("y5_train_glue_f32");
}

extern "C" int y5_raw_to_nhwc_tiled(const void* draw, void* dlogits, int B, int npix, int na, int no, int ld, void* stream_);
extern "C" int y5_nhwc_to_raw(const void* logits, void* raw, int B, int npix, int na, int no, int ld, void* stream_) {
  if (!logits || !raw || B < 1 || npix < 1 || na < 1 || no < 1 || ld < na * no) return y5_fail(Y5_ERR_BAD_ARG, "nhwc_to_raw: bad args");
  if ((ld & 7) == 0 && na <= 8 && (((uintptr_t)logits | (uintptr_t)raw) & 15) == 0) {
    // LDS-tiled path of the inference decode kernel, raw output only (z = NULL): nx = npix, ny = 1 keeps the row math trivial
    const float anchors[16] = {0};
    return y5_detect_decode(logits, Y5_F16, B, 1, npix, na, no, 0, ld, 1.0f, anchors, nullptr, Y5_F16, (long long)na * npix, 0, raw, stream_);
  }
  const long long total = (long long)B * na * npix * no;
  hipLaunchKernelGGL(y5_nhwc_to_raw_kernel, dim3(nblk(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), (const half_t*)logits,
                     (half_t*)raw, npix, na, no, ld, total);
  return y5_check_launch("y5_nhwc_to_raw");
}
extern "C" int y5_raw_to_nhwc(const void* draw, void* dlogits, int B, int npix, int na, int no, int ld, void* stream_) {
  if (!draw || !dlogits || B < 1 || npix < 1 || na < 1 || no < 1 || ld < na * no) return y5_fail(Y5_ERR_BAD_ARG, "raw_to_nhwc: bad args");
  if ((ld & 7) == 0 && (((uintptr_t)draw | (uintptr_t)dlogits) & 15) == 0 && (long long)64 * no < 65536)
    return y5_raw_to_nhwc_tiled(draw, dlogits, B, npix, na, no, ld, stream_);
  const long long total = (long long)B * npix * ld;
  hipLaunchKernelGGL(y5_raw_to_nhwc_kernel, dim3(nblk(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), (const half_t*)draw,
                     (half_t*)dlogits, npix, na, no, ld, total);
  return y5_check_launch("y5_raw_to_nhwc");
}
extern "C" int y5_upsample2x_bwd(const void* gup, void* gsrc, int B, int H, int W, int C, int ld_up, int ld_src, int accumulate, void* stream_) {
  if (!gup || !gsrc || C % 8 || ld_up % 8 || ld_src % 8) return y5_fail(Y5_ERR_BAD_ARG, "upsample2x_bwd: bad args");
  const int vpp = C / 8;
  const long long total = (long long)B * H * W * vpp;
  hipLaunchKernelGGL(y5_upsample2x_bwd_kernel, dim3(nblk(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), (const char*)gup,
                     (char*)gsrc, H, W, vpp, ld_up * 2, ld_src * 2, accumulate, total);
  return y5_check_launch("y5_upsample2x_bwd");
}
extern "C" int y5_add_slice(const void* src, void* dst, long long npix, int C, int lds, int ldd, int accumulate, void* stream_) {
  if (!src || !dst || C % 8 || lds % 8 || ldd % 8) return y5_fail(Y5_ERR_BAD_ARG, "add_slice: bad args");
  const int vpp = C / 8;
  const long long total = npix * vpp;
  hipLaunchKernelGGL(y5_add_slice_kernel, dim3(nblk(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), (const char*)src,
                     (char*)dst, vpp, lds * 2, ldd * 2, accumulate, total);
  return y5_check_launch("y5_add_slice");
}
extern "C" int y5_sppf_pool_bwd(const void* act, void* grad, int B, int H, int W, int C, int ld_act, int ld_grad, int k, void* stream_) {
  if (!act || !grad || C % 8 || ld_act % 8 || ld_grad % 8 || ld_act < 4 * C || ld_grad < 4 * C || !(k & 1)) return y5_fail(Y5_ERR_BAD_ARG, "sppf_pool_bwd: bad args");
  // 16-byte channel groups per workgroup: 2 = whole 32-byte sectors per pixel access, but only while two workgroups still fit a CU (the kernel is a chain
  // of load -> barrier -> scatter -> barrier phases: one big workgroup per CU sat through every memory latency alone); Y5_SPPF_BWD_GV forces 1 / 2 / 4
  static const int force_gv = getenv("Y5_SPPF_BWD_GV") ? atoi(getenv("Y5_SPPF_BWD_GV")) : 0;
  static const int force_gather = getenv("Y5_SPPF_BWD_GATHER") ? atoi(getenv("Y5_SPPF_BWD_GATHER")) : 0;
  if (k > 15) return y5_fail(Y5_ERR_UNSUPPORTED, "sppf_pool_bwd: kernel size above 15");
  constexpr size_t kFix = 16 + 32 + 64, kGather = 16 + 32 + 32 + 8, kMax = 150 * 1024;   // LDS bytes per vector: a_in, g_out, g_in (, amax)
  const size_t hw = (size_t)H * W;
  const bool gather = force_gather || hw * kFix > kMax;
  const size_t per = gather ? kGather : kFix;
  if (hw * per > kMax) return y5_fail(Y5_ERR_UNSUPPORTED, "sppf_pool_bwd: H*W plane does not fit in LDS");
  int gv = (C % 16 == 0 && hw * 2 * per <= 75 * 1024) ? 2 : 1;
  if ((force_gv == 2 || force_gv == 4) && C % (8 * force_gv) == 0 && hw * force_gv * per <= kMax) gv = force_gv;
  if (force_gv == 1) gv = 1;
  const size_t lds = hw * gv * per;
#ifndef Y5_EMU
  {   //  the plane sizes above assume the 160 KiB LDS of gfx950: ask the device instead of launching what cannot start
    int dev = 0, cap = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && cap > 0 && lds > (size_t)cap)
      return y5_fail(Y5_ERR_UNSUPPORTED, "sppf_pool_bwd: H*W plane does not fit in this device's LDS");
  }
#endif
  static bool a = false;
  if (!a) {
    hipFuncSetAttribute((const void*)y5_sppf_pool_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)y5_sppf_pool_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)y5_sppf_pool_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)y5_sppf_pool_bwd_gather_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)y5_sppf_pool_bwd_gather_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)y5_sppf_pool_bwd_gather_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    a = true;
  }
  const dim3 g((unsigned)(B * (C / (8 * gv))));
  hipStream_t st = static_cast<hipStream_t>(stream_);
#define Y5_PB_LAUNCH(KERN) hipLaunchKernelGGL(KERN, g, dim3(256), lds, st, (const char*)act, (char*)grad, H, W, C * 2, ld_act * 2, ld_grad * 2, k)
  if (gather) {
    if (gv == 2) Y5_PB_LAUNCH(y5_sppf_pool_bwd_gather_kernel<2>);
    else if (gv == 4) Y5_PB_LAUNCH(y5_sppf_pool_bwd_gather_kernel<4>);
    else Y5_PB_LAUNCH(y5_sppf_pool_bwd_gather_kernel<1>);
  } else {
    if (gv == 2) Y5_PB_LAUNCH(y5_sppf_pool_bwd_kernel<2>);
    else if (gv == 4) Y5_PB_LAUNCH(y5_sppf_pool_bwd_kernel<4>);
    else Y5_PB_LAUNCH(y5_sppf_pool_bwd_kernel<1>);
  }
#undef Y5_PB_LAUNCH
  return y5_check_launch("y5_sppf_pool_bwd");
}

// ---- filter (re)packing on the device: fp32 master weights -> the fp16 layouts the kernels stream ---------------------
// (replaces a dozen torch micro-ops per convolution per step; weights change at every optimizer step)
// forward: out[n][k = (kh*KW + kw)*C1v + c] = w[n][c][kh][kw]   (c < C1, n < C2; everything else zero)
__global__ void y5_pack_conv_weight_kernel(const float* __restrict__ w, half_t* __restrict__ out, int C2, int C1, int KH, int KW, int C1v,
                                           int Kpad, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % Kpad);
  const int n = (int)(i / Kpad);
  float v = 0.f;
  const int K = KH * KW * C1v;
  if (n < C2 && k < K) {
    const int c = k % C1v, t = k / C1v;
    const int kh = t / KW, kw = t - kh * KW;
    if (c < C1) v = w[((n * C1 + c) * KH + kh) * KW + kw];
  }
  out[i] = (half_t)v;
}
// data-gradient sub-filter of one parity class: out[n = c1][k = (a*NTW + b)*C2v + c2] = w[c2][c1][th[a]][tw[b]]
struct Y5DgradTaps { int th[8], tw[8]; };
__global__ void y5_pack_dgrad_weight_kernel(const float* __restrict__ w, half_t* __restrict__ out, int C2, int C1, int KH, int KW, int NTH,
                                            int NTW, Y5DgradTaps taps, int C2v, int Kpad, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % Kpad);
  const int n = (int)(i / Kpad);
  float v = 0.f;
  const int K = NTH * NTW * C2v;
  if (n < C1 && k < K) {
    const int c2 = k % C2v, t = k / C2v;
    const int a = t / NTW, b = t - a * NTW;
    if (c2 < C2) v = w[((c2 * C1 + n) * KH + taps.th[a]) * KW + taps.tw[b]];
  }
  out[i] = (half_t)v;
}
// weight gradient back to the parameter layout: gw[n][c][kh][kw] = dw[n][(kh*KW + kw)*C1v + c]
__global__ void y5_unpack_conv_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ gw, int C1, int KH, int KW, int C1v, int Kpad,
                                            long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over gw elements
  if (i >= total) return;
  const int kw = (int)(i % KW);
  long long t = i / KW;
  const int kh = (int)(t % KH);
  t /= KH;
  const int c = (int)(t % C1);
  const long long n = t / C1;
  gw[i] = dw[n * Kpad + (kh * KW + kw) * C1v + c];
}

extern "C" int y5_pack_conv_weight(const float* w, int C2, int C1, int KH, int KW, int C1_view, void* out_f16, int Kpad, int Npad, void* stream_) {
  if (!w || !out_f16 || C1_view < C1 || Kpad < KH * KW * C1_view || Npad < C2) return y5_fail(Y5_ERR_BAD_ARG, "pack_conv_weight: bad args");
  const long long total = (long long)Npad * Kpad;
  hipLaunchKernelGGL(y5_pack_conv_weight_kernel, dim3(nblk(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), w, (half_t*)out_f16, C2, C1,
                     KH, KW, C1_view, Kpad, total);
  return y5_check_launch("y5_pack_conv_weight");
}
extern "C" int y5_pack_dgrad_weight(const float* w, int C2, int C1, int KH, int KW, const int* taps_h, int nth, const int* taps_w, int ntw,
                                    int C2_view, void* out_f16, int Kpad, int Npad, void* stream_) {
  if (!w || !out_f16 || !taps_h || !taps_w || nth < 1 || nth > 8 || ntw < 1 || ntw > 8 || C2_view < C2 || Kpad < nth * ntw * C2_view || Npad < C1)
    return y5_fail(Y5_ERR_BAD_ARG, "pack_dgrad_weight: bad args");
  Y5DgradTaps t{};
  for (int i = 0; i < nth; ++i) t.th[i] = taps_h[i];
  for (int i = 0; i < ntw; ++i) t.tw[i] = taps_w[i];
  const long long total = (long long)Npad * Kpad;
  hipLaunchKernelGGL(y5_pack_dgrad_weight_kernel, dim3(nblk(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), w, (half_t*)out_f16, C2, C1,
                     KH, KW, nth, ntw, t, C2_view, Kpad, total);
  return y5_check_launch("y5_pack_dgrad_weight");
}
extern "C" int y5_unpack_conv_wgrad(const float* dw_packed, int Kpad, float* gw, int C2, int C1, int KH, int KW, int C1_view, void* stream_) {
  if (!dw_packed || !gw || C1_view < C1 || Kpad < KH * KW * C1_view) return y5_fail(Y5_ERR_BAD_ARG, "unpack_conv_wgrad: bad args");
  const long long total = (long long)C2 * C1 * KH * KW;
  hipLaunchKernelGGL(y5_unpack_conv_wgrad_kernel, dim3(nblk(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), dw_packed, gw, C1, KH, KW,
                     C1_view, Kpad, total);
  return y5_check_launch("y5_unpack_conv_wgrad");
}
// ---- the three transforms above over MANY filters in one launch (one job per filter, table in device memory) ---------------
// A training step re-packs ~57 forward filters and ~77 data-gradient sub-filters and unpacks ~60 weight gradients: 3 launches instead of ~194.
namespace { constexpr int MT_CH = 1024; }
// Persistent workgroups walk the global list of 1024-element chunks (job after job): chunk c -> (job, chunk of the job) by advancing a running
// (job, first chunk of the job) pair -- every workgroup does real work and a thread touches 4 elements per chunk.  (The first version launched
// (chunks of the LARGEST filter) x (jobs) workgroups of 4096 elements: most exited at once, the rest ran 16 dependent gather iterations per thread --
// 137 us per launch for 20-30 MB of traffic.)
__global__ __launch_bounds__(256)
void y5_filter_jobs_kernel(const y5_filter_job* __restrict__ jobs, int njobs) {
  int job = 0;
  long long first = 0;
  long long nch = (jobs[0].total + MT_CH - 1) / MT_CH;
  for (long long ch = blockIdx.x;; ch += gridDim.x) {
    while (job < njobs && ch >= first + nch) {
      first += nch;
      if (++job < njobs) nch = (jobs[job].total + MT_CH - 1) / MT_CH;
    }
    if (job >= njobs) break;
    const y5_filter_job& j = jobs[job];  // (read in place: a by-value copy that changes per chunk lands in scratch memory)
    const long long base = (ch - first) * MT_CH;
    const long long end = base + MT_CH < j.total ? base + MT_CH : j.total;
  // element indices fit 32 bits (y5_filter_jobs checks): unsigned 32-bit divisions instead of the ~100-instruction 64-bit sequences
  for (unsigned i = (unsigned)base + threadIdx.x; i < (unsigned)end; i += 256) {
    if (j.kind == 3) {  // stem filter (C2, 3, 6, 6) fp32 -> [Npad][144] fp16 of y5_conv_stem_fwd: k = (c*6 + kh)*8 + kw, taps kw = 6, 7 zero
      const unsigned k = i % 144u, n = i / 144u;
      const unsigned kw = k & 7u, kh = (k >> 3) % 6u, c = k / 48u;
      float v = 0.f;
      if ((int)n < j.C2 && kw < 6u) v = static_cast<const float*>(j.src)[((n * 3u + c) * 6u + kh) * 6u + kw];
      static_cast<half_t*>(j.dst)[i] = (half_t)v;
      continue;
    }
    if (j.kind == 2) {  // packed fp32 dW -> (C2, C1, KH, KW)
      const unsigned kw = i % (unsigned)j.KW;
      unsigned t = i / (unsigned)j.KW;
      const unsigned kh = t % (unsigned)j.KH;
      t /= (unsigned)j.KH;
      const unsigned c = t % (unsigned)j.C1;
      const unsigned n = t / (unsigned)j.C1;
      static_cast<float*>(j.dst)[i] = static_cast<const float*>(j.src)[(size_t)n * j.Kpad + (kh * j.KW + kw) * j.C1_view + c];
      continue;
    }
    const float* w = static_cast<const float*>(j.src);
    const int k = (int)(i % (unsigned)j.Kpad);
    const int n = (int)(i / (unsigned)j.Kpad);
    float v = 0.f;
    if (j.kind == 0) {  // forward filter
      const int K = j.KH * j.KW * j.C1_view;
      if (n < j.C2 && k < K) {
        const int c = k % j.C1_view, t = k / j.C1_view;
        const int kh = t / j.KW, kw = t - kh * j.KW;
        if (c < j.C1) v = w[((n * j.C1 + c) * j.KH + kh) * j.KW + kw];
      }
    } else {            // data-gradient sub-filter of one parity class
      const int K = j.nth * j.ntw * j.C2_view;
      if (n < j.C1 && k < K) {
        const int c2 = k % j.C2_view, t = k / j.C2_view;
        const int a = t / j.ntw, b = t - a * j.ntw;
        if (c2 < j.C2) v = w[((c2 * j.C1 + n) * j.KH + j.th[a]) * j.KW + j.tw[b]];
      }
    }
    if (j.reserved == 1) static_cast<float*>(j.dst)[i] = v;  // fp32 training plan (reference-precision mode)
    else static_cast<half_t*>(j.dst)[i] = (half_t)v;
  }
  }
}

extern "C" int y5_filter_jobs(const y5_filter_job* jobs_dev, int njobs, long long max_total, void* stream_) {
  if (!jobs_dev || njobs < 1 || njobs > 65535 || max_total < 1) return y5_fail(Y5_ERR_BAD_ARG, "filter_jobs: bad args");
  if (max_total >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "filter_jobs: filter exceeds 2^31 elements");
  // at most njobs * ceil(max_total / chunk) chunks exist; eight workgroups per CU walk them
  const long long upper = (long long)njobs * ((max_total + MT_CH - 1) / MT_CH);
  const long long grid = upper < 2048 ? upper : 2048;
  hipLaunchKernelGGL(y5_filter_jobs_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream_), jobs_dev, njobs);
  return y5_check_launch("y5_filter_jobs");
}
extern "C" int y5_memset_zero(void* p, size_t bytes, void* stream_) {
  if (!p) return y5_fail(Y5_ERR_BAD_ARG, "memset_zero: null");
  if (hipMemsetAsync(p, 0, bytes, static_cast<hipStream_t>(stream_)) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "memset_zero failed");
  return Y5_OK;
}
