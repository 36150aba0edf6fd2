// Train-mode BatchNorm + SiLU around the convolution (models/common.py:82-88 `Conv.forward`: act(bn(conv(x))), BN with
// batch statistics, eps = 1e-3 / momentum = 0.03 from initialize_weights, models/yolo.py:259) and its backward.
// All tensors are NHWC channel slices (pixel stride ld); C % 8 == 0 (fp16) / C % 4 == 0 (fp32): 16 bytes per lane.
//
//   y5_chan_reduce_kernel<MODE>   per-channel sums over all pixels, deterministic: per-block partials, fixed-order finish
//       MODE 0  s0 = sum z            s1 = sum z^2                       (batch statistics)
//       MODE 1  s0 = sum dv           s1 = sum dv * zhat                 (dbeta, dgamma;  dv = dy * silu'(gamma*zhat+beta))
//       MODE 2  s0 = sum dz                                              (bias gradient of a bias-only conv: Detect.m[i])
//   y5_bn_finish_kernel           partials -> mean / invstd (+ running stats update)   or   -> dgamma / dbeta
//   y5_bn_silu_apply_kernel       y = [res +] silu(gamma * (z - mean) * invstd + beta)
//   y5_bn_silu_bwd_apply_kernel   dz = gamma * invstd * (dv - dbeta/N - zhat * dgamma/N)
#pragma once
#include "y5_common.h"

struct Y5BnParams {
  const void* z;        // conv output (pre-BN), NHWC slice
  const void* dy;       // gradient w.r.t. the block output (backward)
  const void* res;      // optional residual added after the activation (forward apply)
  void* out;            // y (forward apply) / dz (backward apply)
  const float* gamma; const float* beta;
  float* mean; float* invstd;           // [C] saved batch statistics
  float* running_mean; float* running_var;  // optional, updated in place by the stats finish
  float* partial;       // [nblk][2][C]
  float* dgamma; float* dbeta;          // [C]
  double* sums;         // [2][C] raw sums (MODE 3 finish / y5_bn_from_sums_kernel): what SyncBatchNorm exchanges between ranks
  long long count;      // pixels the statistics cover (npix of this rank, or of all ranks after the exchange)
  long long npix;
  int C, ldz, ldy, ldr, ldo;
  int nblk;
  float eps, momentum;
  int rev;              // statistics passes walk the tensor from its END: the producer (conv / dgrad) wrote it front to back, so the tail is what the
                        // memory-side cache still holds; the apply pass that follows runs front to back again and meets what this pass read last
};

template <typename T> struct Y5Vec;
template <> struct Y5Vec<half_t> { typedef half8_t V; static constexpr int N = 8; };
template <> struct Y5Vec<float> { typedef float4_t V; static constexpr int N = 4; };

// d/dv [v * sigmoid(v)].  FAST (fp16 activations): v_rcp_f32 (1 ulp) instead of an IEEE division (a 10-instruction sequence) -- the backward
// statistics pass runs this once per activation element and was VALU-bound on it (7.2 GB in 1.93 ms = 3.7 TB/s against the 5.4 TB/s of the forward
// statistics pass).  The fp32 plan keeps the division: its 24-step training run is compared with the oracle's autograd to 1e-3 in the PARAMETERS,
// and the approximate reciprocal alone moved that from 1e-4 to 2.8e-3 (tests/test_gpu_loops.py).
template <bool FAST>
__device__ __forceinline__ float y5_silu_grad(float v) {
  const float d = 1.0f + __expf(-v);
  const float s = FAST ? __builtin_amdgcn_rcpf(d) : 1.0f / d;
  return s * (1.0f + v * (1.0f - s));
}

template <typename T, int MODE>
__global__ __launch_bounds__(256)
void y5_chan_reduce_kernel(const Y5BnParams p) {
  typedef typename Y5Vec<T>::V V;
  constexpr int N = Y5Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_red = reinterpret_cast<float*>(smem);  // [rows][lanes_c][2*N]
  const int lanes_c = p.C / N;
  const int rows = 256 / lanes_c;
  const int tid = threadIdx.x;
  const int r = tid / lanes_c, cl = tid - r * lanes_c;
  float a0[N], a1[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
  if (r < rows) {
    float mu[N], is[N], ga[N], be[N];
    if constexpr (MODE == 1) {
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const int c = cl * N + e;
        mu[e] = p.mean[c]; is[e] = p.invstd[c]; ga[e] = p.gamma[c]; be[e] = p.beta[c];
      }
    }
    // pixels of this block: a contiguous range split evenly over the grid, rows of the block stride through it
    const long long per = (p.npix + gridDim.x - 1) / gridDim.x;
    const long long bx = p.rev ? (long long)gridDim.x - 1 - blockIdx.x : blockIdx.x;   // (partials stay indexed by blockIdx: the finish order is fixed either way)
    const long long p0 = per * bx, p1 = p0 + per < p.npix ? p0 + per : p.npix;
    // U independent 16-byte loads per thread and iteration: the pass is a pure stream, its speed is the number of bytes in flight
    constexpr int U = 4;
    auto accumulate = [&](const V& zv, const V& gv) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int e = 0; e < N; ++e) { const float z = (float)zv[e]; a0[e] += z; a1[e] += z * z; }
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float zh = ((float)zv[e] - mu[e]) * is[e];
          const float dv = (float)gv[e] * y5_silu_grad<sizeof(T) == 2>(ga[e] * zh + be[e]);
          a0[e] += dv; a1[e] += dv * zh;
        }
      } else {
#pragma unroll
        for (int e = 0; e < N; ++e) a0[e] += (float)gv[e];
      }
    };
    const T* zb = static_cast<const T*>(p.z);
    const T* gb = static_cast<const T*>(p.dy);
    long long px = p0 + r;
    for (; px + (long long)(U - 1) * rows < p1; px += (long long)U * rows) {
      V zv[U] = {}, gv[U] = {};
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long q = px + (long long)u * rows;
        if constexpr (MODE != 2) zv[u] = *reinterpret_cast<const V*>(zb + q * p.ldz + cl * N);
        if constexpr (MODE != 0) gv[u] = *reinterpret_cast<const V*>(gb + q * p.ldy + cl * N);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(zv[u], gv[u]);
    }
    for (; px < p1; px += rows) {
      V zv{}, gv{};
      if constexpr (MODE != 2) zv = *reinterpret_cast<const V*>(zb + px * p.ldz + cl * N);
      if constexpr (MODE != 0) gv = *reinterpret_cast<const V*>(gb + px * p.ldy + cl * N);
      accumulate(zv, gv);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
      s_red[(r * lanes_c + cl) * 2 * N + e] = a0[e];
      s_red[(r * lanes_c + cl) * 2 * N + N + e] = a1[e];
    }
  }
  __syncthreads();
  // fixed-order sum over the block's rows: thread (cl, which, e) walks r = 0..rows-1
  const int nout = lanes_c * 2 * N;
  for (int o = tid; o < nout; o += 256) {
    const int c2 = o / (2 * N), k = o - c2 * 2 * N;
    float s = 0.f;
    for (int rr = 0; rr < rows; ++rr) s += s_red[(rr * lanes_c + c2) * 2 * N + k];
    const int which = k / N, e = k - which * N;
    p.partial[((long long)blockIdx.x * 2 + which) * p.C + c2 * N + e] = s;
  }
}

// One workgroup per channel: 128 threads sum the per-block partials (strided, fp64), fixed-order tree in LDS.
// MODE 0: batch statistics (+ running update, torch semantics: running_var uses the unbiased estimate)
// MODE 1/2: plain sums -> dbeta (s0), dgamma (s1)
template <int MODE>
__global__ __launch_bounds__(128)
void y5_bn_finish_kernel(const Y5BnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* s_red = reinterpret_cast<double*>(smem);  // [2][128]
  const int c = blockIdx.x;
  const int tid = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int b = tid; b < p.nblk; b += 128) {
    s0 += (double)p.partial[((long long)b * 2 + 0) * p.C + c];
    s1 += (double)p.partial[((long long)b * 2 + 1) * p.C + c];
  }
  s_red[tid] = s0;
  s_red[128 + tid] = s1;
  __syncthreads();
  for (int d = 64; d > 0; d >>= 1) {
    if (tid < d) { s_red[tid] += s_red[tid + d]; s_red[128 + tid] += s_red[128 + tid + d]; }
    __syncthreads();
  }
  if (tid != 0) return;
  s0 = s_red[0]; s1 = s_red[128];
  if constexpr (MODE == 3) {   // raw fp64 sums: the cross-rank exchange of SyncBatchNorm adds them before mean / invstd exist
    p.sums[c] = s0;
    p.sums[p.C + c] = s1;
  } else if constexpr (MODE == 0) {
    const double n = (double)p.npix;
    const double mean = s0 / n;
    double var = s1 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    p.mean[c] = (float)mean;
    p.invstd[c] = (float)(1.0 / sqrt(var + (double)p.eps));
    if (p.running_mean) p.running_mean[c] = (1.0f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean;
    if (p.running_var) p.running_var[c] = (1.0f - p.momentum) * p.running_var[c] + p.momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
  } else {
    if (p.dbeta) p.dbeta[c] = (float)s0;
    if (p.dgamma) p.dgamma[c] = (float)s1;
  }
}

// Batch statistics from (possibly cross-rank) sums over p.count pixels: torch.nn.SyncBatchNorm's forward (train.py:269-271 converts the model) reduces
// to this once sum z and sum z^2 of all ranks have been added; running statistics use the GLOBAL count (unbiased variance), as there.
__global__ __launch_bounds__(256)
void y5_bn_from_sums_kernel(const Y5BnParams p) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.C) return;
  const double n = (double)p.count;
  const double mean = p.sums[c] / n;
  double var = p.sums[p.C + c] / n - mean * mean;
  if (var < 0.0) var = 0.0;
  p.mean[c] = (float)mean;
  p.invstd[c] = (float)(1.0 / sqrt(var + (double)p.eps));
  if (p.running_mean) p.running_mean[c] = (1.0f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean;
  if (p.running_var) p.running_var[c] = (1.0f - p.momentum) * p.running_var[c] + p.momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
}

// Thread mapping of the two apply kernels (as y5_chan_reduce_kernel): thread = (pixel row r of the block, channel vector cl) with cl FIXED for the
// life of the thread, so the per-channel constants are loaded once into registers and a pixel costs no integer division -- the first version
// decomposed a flat 64-bit vector index per iteration (a ~100-instruction division) and re-loaded 4-6 per-channel values per ELEMENT.
template <typename T, bool RES>
__global__ __launch_bounds__(256)
void y5_bn_silu_apply_kernel(const Y5BnParams p) {
  typedef typename Y5Vec<T>::V V;
  constexpr int N = Y5Vec<T>::N;
  constexpr bool FAST = sizeof(T) == 2;   // fp16 activations: v_rcp_f32 in the sigmoid; the fp32 plan keeps the IEEE division (see y5_silu_grad)
  const int lanes_c = p.C / N;
  const int rows = 256 / lanes_c;
  const int r = threadIdx.x / lanes_c, cl = threadIdx.x - r * lanes_c;
  if (r >= rows) return;
  float mu[N], sc[N], be[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const int c = cl * N + e;
    mu[e] = p.mean[c]; sc[e] = p.gamma[c] * p.invstd[c]; be[e] = p.beta[c];
  }
  const T* zb = static_cast<const T*>(p.z) + cl * N;
  const T* rb = static_cast<const T*>(p.res) + cl * N;
  T* ob = static_cast<T*>(p.out) + cl * N;
  for (long long px = (long long)blockIdx.x * rows + r; px < p.npix; px += (long long)gridDim.x * rows) {
    const V zv = *reinterpret_cast<const V*>(zb + px * p.ldz);
    V rv;
    if constexpr (RES) rv = *reinterpret_cast<const V*>(rb + px * p.ldr);
    V o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float u = ((float)zv[e] - mu[e]) * sc[e] + be[e];
      const float d = 1.0f + __expf(-u);
      float y = FAST ? u * __builtin_amdgcn_rcpf(d) : u / d;
      if constexpr (RES) y += (float)rv[e];
      o[e] = (T)y;
    }
    *reinterpret_cast<V*>(ob + px * p.ldo) = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256)
void y5_bn_silu_bwd_apply_kernel(const Y5BnParams p) {
  typedef typename Y5Vec<T>::V V;
  constexpr int N = Y5Vec<T>::N;
  const int lanes_c = p.C / N;
  const int rows = 256 / lanes_c;
  const int r = threadIdx.x / lanes_c, cl = threadIdx.x - r * lanes_c;
  if (r >= rows) return;
  const float inv_n = 1.0f / (float)(p.count > 0 ? p.count : p.npix);   // SyncBatchNorm: the sums and the count are those of all ranks
  float mu[N], is[N], ga[N], be[N], db[N], dg[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const int c = cl * N + e;
    mu[e] = p.mean[c]; is[e] = p.invstd[c]; ga[e] = p.gamma[c]; be[e] = p.beta[c]; db[e] = p.dbeta[c] * inv_n; dg[e] = p.dgamma[c];
  }
  const T* zb = static_cast<const T*>(p.z) + cl * N;
  const T* gb = static_cast<const T*>(p.dy) + cl * N;
  T* ob = static_cast<T*>(p.out) + cl * N;
  for (long long px = (long long)blockIdx.x * rows + r; px < p.npix; px += (long long)gridDim.x * rows) {
    const V zv = *reinterpret_cast<const V*>(zb + px * p.ldz);
    const V gv = *reinterpret_cast<const V*>(gb + px * p.ldy);
    V o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float zh = ((float)zv[e] - mu[e]) * is[e];
      const float dv = (float)gv[e] * y5_silu_grad<sizeof(T) == 2>(ga[e] * zh + be[e]);
      o[e] = (T)(ga[e] * is[e] * (dv - db[e] - zh * dg[e] * inv_n));
    }
    *reinterpret_cast<V*>(ob + px * p.ldo) = o;
  }
}
