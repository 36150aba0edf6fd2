// Fused optimizer step over all parameter tensors of the model (train.py:413-421):
//     scaler.unscale_(optimizer)                               g *= inv_scale, found_inf = any non-finite g
//     torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0) g *= min(1, max_norm / (||g||_2 + 1e-6))
//     scaler.step(optimizer)                                   SGD(momentum, nesterov) over the 3 groups of
//                                                              utils/torch_utils.py:257-290 (skipped when found_inf)
//     ema.update(model)                                        e = d * e + (1 - d) * p   (utils/torch_utils.py:354-365)
// 177 tensors (yolov5s) would be ~1000 elementwise launches in stock torch; here: one deterministic sum-of-squares pass, one
// 1-block finish and one update pass, all driven by a device-resident tensor table.  Grid = (chunks of CH elements, tensors);
// workgroups past a tensor's end exit at once.  Built with -ffp-contract=off: every product / sum is rounded as torch's
// separate mul / add kernels round them.
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "y5_common.h"
#include "y5_host.h"

namespace {
constexpr int CH = 16384;  // elements per workgroup: 256 threads x 16 float4

struct MtStats { float total_norm, clip_coef, found_inf, pad; };
}  // namespace

__global__ __launch_bounds__(256)
void y5_mt_sumsq_kernel(const y5_mt_tensor* __restrict__ tab, double* __restrict__ partial, int* __restrict__ nonfinite, int max_chunks) {
  const y5_mt_tensor t = tab[blockIdx.y];
  const long long base = (long long)blockIdx.x * CH;
  double* out = partial + (size_t)blockIdx.y * max_chunks + blockIdx.x;
  if (base >= t.n) {
    if (threadIdx.x == 0) *out = 0.0;
    return;
  }
  const float* g = static_cast<const float*>(t.grad);
  const long long end = base + CH < t.n ? base + CH : t.n;
  float acc = 0.f;
  bool bad = false;
  for (long long i = base + threadIdx.x; i < end; i += 256) {
    const float v = g[i];
    bad |= !(fabsf(v) <= 3.402823466e38f);  // inf or nan
    acc += v * v;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_red = reinterpret_cast<float*>(smem);
  int& s_bad = *reinterpret_cast<int*>(smem + 256 * sizeof(float));
  if (threadIdx.x == 0) s_bad = 0;
  s_red[threadIdx.x] = acc;
  __syncthreads();
  if (bad) s_bad = 1;
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *out = (double)s_red[0];
    if (s_bad) atomicExch(nonfinite, 1);
  }
}

__global__ __launch_bounds__(256)
void y5_mt_norm_finish_kernel(const double* __restrict__ partial, int n, const int* __restrict__ nonfinite, float inv_scale, float max_norm,
                              MtStats* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* s_red = reinterpret_cast<double*>(smem);
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s_red[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)(sqrt(s_red[0]) * (double)inv_scale);  // norm of the UNSCALED gradients
    float coef = 1.0f;
    if (max_norm > 0.f) {
      coef = max_norm / (total + 1e-6f);
      if (!(coef < 1.0f)) coef = 1.0f;  // torch.clamp(max=1.0); a NaN norm leaves the step to found_inf
    }
    stats->total_norm = total;
    stats->clip_coef = coef;
    stats->found_inf = (*nonfinite != 0 || !(total <= 3.402823466e38f)) ? 1.0f : 0.0f;
  }
}

// groups: lr[g], weight_decay[g] for g = tensor.group (0..3)
struct MtHyper { float lr[4]; float wd[4]; float momentum, inv_scale, ema_d; int nesterov, use_stats; };

__global__ __launch_bounds__(256)
void y5_mt_sgd_kernel(const y5_mt_tensor* __restrict__ tab, const MtStats* __restrict__ stats, const MtHyper h) {
  const y5_mt_tensor t = tab[blockIdx.y];
  const long long base = (long long)blockIdx.x * CH;
  if (base >= t.n) return;
  float coef = 1.0f;
  bool skip = false;  // GradScaler.step: a non-finite gradient skips the parameter update (the EMA still moves, train.py:417-421)
  if (h.use_stats) {
    skip = stats->found_inf != 0.f;
    coef = stats->clip_coef;
  }
  const long long end = base + CH < t.n ? base + CH : t.n;
  float* p = static_cast<float*>(t.param);
  const float* g = static_cast<const float*>(t.grad);
  float* m = static_cast<float*>(t.mom);
  float* e = static_cast<float*>(t.ema);
  const float lr = h.lr[t.group & 3], wd = h.wd[t.group & 3];
  const float one_minus_d = 1.0f - h.ema_d;
  if (skip && !e) return;
  for (long long i = base + threadIdx.x; i < end; i += 256) {
    float w = p[i];
    if (!skip) {
      float d_p = g[i];
      if (h.inv_scale != 1.0f) d_p = d_p * h.inv_scale;
      if (coef != 1.0f) d_p = d_p * coef;
      if (wd != 0.f) d_p = d_p + wd * w;
      if (m) {
        const float b = m[i] * h.momentum + d_p;  // zero-initialised buffer: the first step gives b = d_p exactly (torch clones d_p)
        m[i] = b;
        d_p = h.nesterov ? d_p + h.momentum * b : b;
      }
      w = w - lr * d_p;
      p[i] = w;
    }
    if (e) e[i] = e[i] * h.ema_d + one_minus_d * w;
  }
}

// dst = dst * d + (1 - d) * src  over a table whose entries use (param = src, ema = dst): ModelEMA over float BUFFERS
__global__ __launch_bounds__(256)
void y5_mt_lerp_kernel(const y5_mt_tensor* __restrict__ tab, float d) {
  const y5_mt_tensor t = tab[blockIdx.y];
  const long long base = (long long)blockIdx.x * CH;
  if (base >= t.n) return;
  const long long end = base + CH < t.n ? base + CH : t.n;
  const float* src = static_cast<const float*>(t.param);
  float* dst = static_cast<float*>(t.ema);
  const float omd = 1.0f - d;
  for (long long i = base + threadIdx.x; i < end; i += 256) dst[i] = dst[i] * d + omd * src[i];
}

static int mt_grid(int ntensors, long long max_numel, dim3* grid) {
  if (ntensors < 1 || ntensors > 65535 || max_numel < 1) return y5_fail(Y5_ERR_BAD_ARG, "multi-tensor op: bad tensor count / size");
  const long long chunks = (max_numel + CH - 1) / CH;
  if (chunks > 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "multi-tensor op: tensor too large");
  *grid = dim3((unsigned)chunks, (unsigned)ntensors);
  return Y5_OK;
}

extern "C" size_t y5_mt_workspace_bytes(int ntensors, long long max_numel) {
  const long long chunks = (max_numel + CH - 1) / CH;
  return (size_t)ntensors * (size_t)chunks * sizeof(double) + 64;
}

extern "C" int y5_mt_grad_norm(const y5_mt_tensor* table_dev, int ntensors, long long max_numel, float inv_scale, float max_norm,
                               float* stats_dev, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!table_dev || !stats_dev || !workspace) return y5_fail(Y5_ERR_BAD_ARG, "mt_grad_norm: null pointer");
  dim3 grid;
  if (int rc = mt_grid(ntensors, max_numel, &grid)) return rc;
  if (workspace_bytes < y5_mt_workspace_bytes(ntensors, max_numel)) return y5_fail(Y5_ERR_BAD_ARG, "mt_grad_norm: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  int* flag = static_cast<int*>(workspace);
  double* partial = reinterpret_cast<double*>(static_cast<char*>(workspace) + 64);
  if (hipMemsetAsync(flag, 0, 64, st) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "mt_grad_norm: memset failed");
  hipLaunchKernelGGL(y5_mt_sumsq_kernel, grid, dim3(256), 256 * sizeof(float) + 16, st, table_dev, partial, flag, (int)grid.x);
  hipLaunchKernelGGL(y5_mt_norm_finish_kernel, dim3(1), dim3(256), 256 * sizeof(double), st, partial, (int)(grid.x * grid.y), flag, inv_scale, max_norm,
                     reinterpret_cast<MtStats*>(stats_dev));
  return y5_check_launch("y5_mt_grad_norm");
}

extern "C" int y5_mt_sgd_step(const y5_mt_tensor* table_dev, int ntensors, long long max_numel, const float* lr4, const float* wd4, float momentum,
                              int nesterov, float inv_scale, const float* stats_dev, float ema_decay, void* stream_) {
  if (!table_dev || !lr4 || !wd4) return y5_fail(Y5_ERR_BAD_ARG, "mt_sgd_step: null pointer");
  dim3 grid;
  if (int rc = mt_grid(ntensors, max_numel, &grid)) return rc;
  MtHyper h{};
  for (int i = 0; i < 4; ++i) { h.lr[i] = lr4[i]; h.wd[i] = wd4[i]; }
  h.momentum = momentum; h.inv_scale = inv_scale; h.ema_d = ema_decay; h.nesterov = nesterov;
  h.use_stats = stats_dev != nullptr;
  hipLaunchKernelGGL(y5_mt_sgd_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), table_dev,
                     reinterpret_cast<const MtStats*>(stats_dev), h);
  return y5_check_launch("y5_mt_sgd_step");
}

extern "C" int y5_mt_lerp(const y5_mt_tensor* table_dev, int ntensors, long long max_numel, float decay, void* stream_) {
  if (!table_dev) return y5_fail(Y5_ERR_BAD_ARG, "mt_lerp: null pointer");
  dim3 grid;
  if (int rc = mt_grid(ntensors, max_numel, &grid)) return rc;
  hipLaunchKernelGGL(y5_mt_lerp_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), table_dev, decay);
  return y5_check_launch("y5_mt_lerp");
}
