// C-ABI launchers of the train-mode BatchNorm + SiLU kernels (bn_kernels.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/yolov5_hip.h"
#include "bn_kernels.h"
#include "y5_host.h"

namespace {
int nblk_for(long long npix) {
  long long n = (npix + 63) / 64;
  return (int)(n < 1 ? 1 : n > 1024 ? 1024 : n);
}
int check(int dt, long long npix, int C, const void* ws, size_t ws_bytes) {
  if (dt != Y5_F16 && dt != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "bn: dtype must be Y5_F16 or Y5_F32");
  const int vec = dt == Y5_F16 ? 8 : 4;
  if (npix < 1 || C < vec || C % vec || C / vec > 256) return y5_fail(Y5_ERR_BAD_ARG, "bn: C must be a multiple of 16 bytes, at most 256 vectors");
  if (!ws || ws_bytes < y5_bn_workspace_bytes(C, npix) || ((uintptr_t)ws & 15)) return y5_fail(Y5_ERR_WORKSPACE, "bn: workspace too small or misaligned");
  return Y5_OK;
}
// (the reversed-walk statistics pass of round 5, Y5_BN_REV, measured neutral -- profiles/experiments/r05_bn_reverse_pass.log -- and lost its switch in round 6;
// Y5BnParams.rev stays 0)
template <int MODE>
void reduce(const Y5BnParams& p0, int dt, hipStream_t st) {
  Y5BnParams p = p0;
  p.rev = 0;
  const int vec = dt == Y5_F16 ? 8 : 4;
  const size_t lds = (size_t)256 * 2 * vec * 4;
  if (dt == Y5_F16) hipLaunchKernelGGL((y5_chan_reduce_kernel<half_t, MODE>), dim3((unsigned)p.nblk), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((y5_chan_reduce_kernel<float, MODE>), dim3((unsigned)p.nblk), dim3(256), lds, st, p);
}
unsigned apply_grid(long long npix, int C, int dt) {
  const long long vecs = npix * (C / (dt == Y5_F16 ? 8 : 4));
  long long g = (vecs + 255) / 256;
  return (unsigned)(g > 16384 ? 16384 : g);
}
}  // namespace

extern "C" size_t y5_bn_workspace_bytes(int C, long long npix) { return (size_t)nblk_for(npix) * 2 * (size_t)C * 4; }

extern "C" int y5_bn_silu_fwd(const void* z, int dt, long long npix, int C, int ldz, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                              const void* residual, int ldr, void* y, int ldy, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = check(dt, npix, C, ws, ws_bytes)) return rc;
  if (!z || !gamma || !beta || !save_mean || !save_invstd || !y) return y5_fail(Y5_ERR_BAD_ARG, "bn_silu_fwd: null pointer");
  Y5BnParams p{};
  p.z = z; p.res = residual; p.out = y; p.gamma = gamma; p.beta = beta; p.mean = save_mean; p.invstd = save_invstd;
  p.running_mean = running_mean; p.running_var = running_var; p.partial = static_cast<float*>(ws);
  p.npix = npix; p.C = C; p.ldz = ldz; p.ldr = ldr; p.ldo = ldy; p.nblk = nblk_for(npix); p.eps = eps; p.momentum = momentum;
  reduce<0>(p, dt, st);
  hipLaunchKernelGGL(y5_bn_finish_kernel<0>, dim3((unsigned)C), dim3(128), 2 * 128 * 8, st, p);
  const dim3 g(apply_grid(npix, C, dt));
  if (dt == Y5_F16) {
    if (residual) hipLaunchKernelGGL((y5_bn_silu_apply_kernel<half_t, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((y5_bn_silu_apply_kernel<half_t, false>), g, dim3(256), 0, st, p);
  } else {
    if (residual) hipLaunchKernelGGL((y5_bn_silu_apply_kernel<float, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((y5_bn_silu_apply_kernel<float, false>), g, dim3(256), 0, st, p);
  }
  return y5_check_launch("y5_bn_silu_fwd");
}

extern "C" int y5_bn_silu_bwd(const void* dy, int ld_dy, const void* z, int ldz, int dt, long long npix, int C, const float* gamma,
                              const float* beta, const float* save_mean, const float* save_invstd, void* dz, int ld_dz, float* dgamma,
                              float* dbeta, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = check(dt, npix, C, ws, ws_bytes)) return rc;
  if (!dy || !z || !gamma || !beta || !save_mean || !save_invstd || !dz || !dgamma || !dbeta) return y5_fail(Y5_ERR_BAD_ARG, "bn_silu_bwd: null pointer");
  Y5BnParams p{};
  p.z = z; p.dy = dy; p.out = dz; p.gamma = gamma; p.beta = beta;
  p.mean = const_cast<float*>(save_mean); p.invstd = const_cast<float*>(save_invstd);
  p.partial = static_cast<float*>(ws); p.dgamma = dgamma; p.dbeta = dbeta;
  p.npix = npix; p.C = C; p.ldz = ldz; p.ldy = ld_dy; p.ldo = ld_dz; p.nblk = nblk_for(npix);
  reduce<1>(p, dt, st);
  hipLaunchKernelGGL(y5_bn_finish_kernel<1>, dim3((unsigned)C), dim3(128), 2 * 128 * 8, st, p);
  const dim3 g(apply_grid(npix, C, dt));
  if (dt == Y5_F16) hipLaunchKernelGGL((y5_bn_silu_bwd_apply_kernel<half_t>), g, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((y5_bn_silu_bwd_apply_kernel<float>), g, dim3(256), 0, st, p);
  return y5_check_launch("y5_bn_silu_bwd");
}

// ---- SyncBatchNorm (train.py:269-271 `torch.nn.SyncBatchNorm.convert_sync_batchnorm`): the two fused entries above, cut at the point where the ranks
// exchange their per-channel sums.  The exchange itself (all-reduce SUM of 2 C doubles forward, 2 C floats backward) is the host's: no collective at the C-ABI.
extern "C" int y5_bn_stats(const void* z, int dt, long long npix, int C, int ldz, double* sums, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = check(dt, npix, C, ws, ws_bytes)) return rc;
  if (!z || !sums) return y5_fail(Y5_ERR_BAD_ARG, "bn_stats: null pointer");
  Y5BnParams p{};
  p.z = z; p.partial = static_cast<float*>(ws); p.sums = sums; p.npix = npix; p.C = C; p.ldz = ldz; p.nblk = nblk_for(npix);
  reduce<0>(p, dt, st);
  hipLaunchKernelGGL(y5_bn_finish_kernel<3>, dim3((unsigned)C), dim3(128), 2 * 128 * 8, st, p);
  return y5_check_launch("y5_bn_stats");
}

extern "C" int y5_bn_silu_fwd_from_sums(const void* z, int dt, long long npix, int C, int ldz, const float* gamma, const float* beta, float eps,
                                        float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, const double* sums,
                                        long long count_total, const void* residual, int ldr, void* y, int ldy, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (dt != Y5_F16 && dt != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "bn: dtype must be Y5_F16 or Y5_F32");
  const int vec = dt == Y5_F16 ? 8 : 4;
  if (npix < 1 || count_total < npix || C < vec || C % vec || C / vec > 256) return y5_fail(Y5_ERR_BAD_ARG, "bn_silu_fwd_from_sums: bad npix / count / C");
  if (!z || !gamma || !beta || !save_mean || !save_invstd || !sums || !y) return y5_fail(Y5_ERR_BAD_ARG, "bn_silu_fwd_from_sums: null pointer");
  Y5BnParams p{};
  p.z = z; p.res = residual; p.out = y; p.gamma = gamma; p.beta = beta; p.mean = save_mean; p.invstd = save_invstd;
  p.running_mean = running_mean; p.running_var = running_var; p.sums = const_cast<double*>(sums); p.count = count_total;
  p.npix = npix; p.C = C; p.ldz = ldz; p.ldr = ldr; p.ldo = ldy; p.eps = eps; p.momentum = momentum;
  hipLaunchKernelGGL(y5_bn_from_sums_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, p);
  const dim3 g(apply_grid(npix, C, dt));
  if (dt == Y5_F16) {
    if (residual) hipLaunchKernelGGL((y5_bn_silu_apply_kernel<half_t, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((y5_bn_silu_apply_kernel<half_t, false>), g, dim3(256), 0, st, p);
  } else {
    if (residual) hipLaunchKernelGGL((y5_bn_silu_apply_kernel<float, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((y5_bn_silu_apply_kernel<float, false>), g, dim3(256), 0, st, p);
  }
  return y5_check_launch("y5_bn_silu_fwd_from_sums");
}

// Forward of BatchNorm + SiLU when the convolution that produced z already left the per-workgroup partial sums (y5_conv2d_fwd_stats): the fixed-order
// finish over `rows` partial rows [2][C], then the apply pass -- the statistics pass over z is gone.
extern "C" int y5_bn_silu_fwd_from_partials(const void* z, int dt, long long npix, int C, int ldz, const float* gamma, const float* beta, float eps,
                                            float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                            const float* partial, int rows, const void* residual, int ldr, void* y, int ldy, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (dt != Y5_F16 && dt != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "bn: dtype must be Y5_F16 or Y5_F32");
  const int vec = dt == Y5_F16 ? 8 : 4;
  if (npix < 1 || rows < 1 || C < vec || C % vec || C / vec > 256) return y5_fail(Y5_ERR_BAD_ARG, "bn_silu_fwd_from_partials: bad npix / rows / C");
  if (!z || !gamma || !beta || !save_mean || !save_invstd || !partial || !y) return y5_fail(Y5_ERR_BAD_ARG, "bn_silu_fwd_from_partials: null pointer");
  Y5BnParams p{};
  p.z = z; p.res = residual; p.out = y; p.gamma = gamma; p.beta = beta; p.mean = save_mean; p.invstd = save_invstd;
  p.running_mean = running_mean; p.running_var = running_var; p.partial = const_cast<float*>(partial);
  p.npix = npix; p.C = C; p.ldz = ldz; p.ldr = ldr; p.ldo = ldy; p.nblk = rows; p.eps = eps; p.momentum = momentum;
  hipLaunchKernelGGL(y5_bn_finish_kernel<0>, dim3((unsigned)C), dim3(128), 2 * 128 * 8, st, p);
  const dim3 g(apply_grid(npix, C, dt));
  if (dt == Y5_F16) {
    if (residual) hipLaunchKernelGGL((y5_bn_silu_apply_kernel<half_t, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((y5_bn_silu_apply_kernel<half_t, false>), g, dim3(256), 0, st, p);
  } else {
    if (residual) hipLaunchKernelGGL((y5_bn_silu_apply_kernel<float, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((y5_bn_silu_apply_kernel<float, false>), g, dim3(256), 0, st, p);
  }
  return y5_check_launch("y5_bn_silu_fwd_from_partials");
}

extern "C" int y5_bn_bwd_stats(const void* dy, int ld_dy, const void* z, int ldz, int dt, long long npix, int C, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = check(dt, npix, C, ws, ws_bytes)) return rc;
  if (!dy || !z || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta) return y5_fail(Y5_ERR_BAD_ARG, "bn_bwd_stats: null pointer");
  Y5BnParams p{};
  p.z = z; p.dy = dy; p.gamma = gamma; p.beta = beta; p.mean = const_cast<float*>(save_mean); p.invstd = const_cast<float*>(save_invstd);
  p.partial = static_cast<float*>(ws); p.dgamma = dgamma; p.dbeta = dbeta;
  p.npix = npix; p.C = C; p.ldz = ldz; p.ldy = ld_dy; p.nblk = nblk_for(npix);
  reduce<1>(p, dt, st);
  hipLaunchKernelGGL(y5_bn_finish_kernel<1>, dim3((unsigned)C), dim3(128), 2 * 128 * 8, st, p);
  return y5_check_launch("y5_bn_bwd_stats");
}

extern "C" int y5_bn_silu_bwd_from_sums(const void* dy, int ld_dy, const void* z, int ldz, int dt, long long npix, int C, const float* gamma, const float* beta,
                                        const float* save_mean, const float* save_invstd, const float* sum_dgamma, const float* sum_dbeta,
                                        long long count_total, void* dz, int ld_dz, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (dt != Y5_F16 && dt != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "bn: dtype must be Y5_F16 or Y5_F32");
  const int vec = dt == Y5_F16 ? 8 : 4;
  if (npix < 1 || count_total < npix || C < vec || C % vec || C / vec > 256) return y5_fail(Y5_ERR_BAD_ARG, "bn_silu_bwd_from_sums: bad npix / count / C");
  if (!dy || !z || !gamma || !beta || !save_mean || !save_invstd || !sum_dgamma || !sum_dbeta || !dz) return y5_fail(Y5_ERR_BAD_ARG, "bn_silu_bwd_from_sums: null pointer");
  Y5BnParams p{};
  p.z = z; p.dy = dy; p.out = dz; p.gamma = gamma; p.beta = beta; p.mean = const_cast<float*>(save_mean); p.invstd = const_cast<float*>(save_invstd);
  p.dgamma = const_cast<float*>(sum_dgamma); p.dbeta = const_cast<float*>(sum_dbeta); p.count = count_total;
  p.npix = npix; p.C = C; p.ldz = ldz; p.ldy = ld_dy; p.ldo = ld_dz;
  const dim3 g(apply_grid(npix, C, dt));
  if (dt == Y5_F16) hipLaunchKernelGGL((y5_bn_silu_bwd_apply_kernel<half_t>), g, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((y5_bn_silu_bwd_apply_kernel<float>), g, dim3(256), 0, st, p);
  return y5_check_launch("y5_bn_silu_bwd_from_sums");
}

extern "C" int y5_channel_sum(const void* x, int dt, long long npix, int C, int ld, float* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = check(dt, npix, C, ws, ws_bytes)) return rc;
  if (!x || !out) return y5_fail(Y5_ERR_BAD_ARG, "channel_sum: null pointer");
  Y5BnParams p{};
  p.dy = x; p.partial = static_cast<float*>(ws); p.dbeta = out; p.dgamma = nullptr;
  p.npix = npix; p.C = C; p.ldy = ld; p.nblk = nblk_for(npix);
  reduce<2>(p, dt, st);
  hipLaunchKernelGGL(y5_bn_finish_kernel<2>, dim3((unsigned)C), dim3(128), 2 * 128 * 8, st, p);
  return y5_check_launch("y5_channel_sum");
}
