// C-ABI launcher for ComputeLoss forward/backward (loss_kernels.h).  Built with -ffp-contract=off so that the target
// assignment arithmetic (fp32 multiply / divide / compare chains of utils/loss.py:205-243) is not re-associated.
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "loss_kernels.h"
#include "y5_host.h"

namespace {

struct LevelOff { size_t rb, ra, rgj, rgi, rcls, next, tbox, anch, iou, rl_box, rl_cls, G, head, obj_part; };
struct Layout { size_t n_rows, obji; LevelOff lv[Y5_LOSS_MAX_NL]; long long cap, cells[Y5_LOSS_MAX_NL]; size_t head_begin, head_end, total; };

size_t take(size_t& o, size_t bytes) { const size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; }

int validate(const y5_loss_desc* d, int nt) {
  if (!d) return y5_fail(Y5_ERR_BAD_ARG, "loss: null descriptor");
  if (d->dtype != Y5_F16 && d->dtype != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "loss: dtype must be Y5_F16 or Y5_F32");
  if (d->nl < 1 || d->nl > Y5_LOSS_MAX_NL || d->na < 1 || d->na > Y5_LOSS_MAX_NA || d->nc < 1 || d->bs < 1 || nt < 0)
    return y5_fail(Y5_ERR_BAD_ARG, "loss: nl/na/nc/bs/nt out of range");
  for (int i = 0; i < d->nl; ++i)
    if (d->ny[i] < 1 || d->nx[i] < 1 || (long long)d->bs * d->na * d->ny[i] * d->nx[i] >= 0x7fffffffLL)
      return y5_fail(Y5_ERR_BAD_ARG, "loss: bad grid size");
  if (5LL * d->na * nt >= 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "loss: too many targets");
  return Y5_OK;
}

Layout layout(const y5_loss_desc* d, int nt) {
  Layout L{};
  const int no = 5 + d->nc;
  const long long cap = 5LL * d->na * nt;
  const size_t capz = (size_t)(cap > 0 ? cap : 1);
  L.cap = cap;
  size_t o = 0;
  L.n_rows = take(o, sizeof(int) * Y5_LOSS_MAX_NL);
  L.obji = take(o, sizeof(float) * Y5_LOSS_MAX_NL);
  for (int i = 0; i < d->nl; ++i) {
    LevelOff& v = L.lv[i];
    v.rb = take(o, capz * 4); v.ra = take(o, capz * 4); v.rgj = take(o, capz * 4); v.rgi = take(o, capz * 4);
    v.rcls = take(o, capz * 4); v.next = take(o, capz * 4);
    v.tbox = take(o, capz * 16); v.anch = take(o, capz * 8);
    v.iou = take(o, capz * 4); v.rl_box = take(o, capz * 4); v.rl_cls = take(o, capz * 4);
    v.G = take(o, capz * no * 4);
    L.cells[i] = (long long)d->bs * d->na * d->ny[i] * d->nx[i];
    v.obj_part = take(o, (size_t)((L.cells[i] + 255) / 256) * 4);
  }
  L.head_begin = o;  // all head arrays are contiguous: one memset(0xFF) per forward
  for (int i = 0; i < d->nl; ++i) L.lv[i].head = take(o, (size_t)L.cells[i] * 4);
  L.head_end = o;
  L.total = o;
  return L;
}

void fill(Y5LossParams& P, const y5_loss_desc* d, const Layout& L, char* ws, int nt) {
  P.nl = d->nl; P.na = d->na; P.nc = d->nc; P.no = 5 + d->nc; P.bs = d->bs; P.nt = nt;
  P.hyp_box = d->hyp_box; P.hyp_obj = d->hyp_obj; P.hyp_cls = d->hyp_cls; P.cls_pw = d->cls_pw; P.obj_pw = d->obj_pw; P.fl_gamma = d->fl_gamma;
  P.anchor_t = d->anchor_t; P.cp = d->cp; P.cn = d->cn;
  P.n_rows = reinterpret_cast<int*>(ws + L.n_rows);
  P.obji = reinterpret_cast<float*>(ws + L.obji);
  for (int i = 0; i < d->nl; ++i) {
    Y5LossLevel& v = P.lv[i];
    const LevelOff& f = L.lv[i];
    v.ny = d->ny[i]; v.nx = d->nx[i]; v.cells = L.cells[i]; v.cap = L.cap;
    v.rb = (int*)(ws + f.rb); v.ra = (int*)(ws + f.ra); v.rgj = (int*)(ws + f.rgj); v.rgi = (int*)(ws + f.rgi);
    v.rcls = (int*)(ws + f.rcls); v.next = (int*)(ws + f.next);
    v.tbox = (float*)(ws + f.tbox); v.anch = (float*)(ws + f.anch); v.iou = (float*)(ws + f.iou);
    v.rl_box = (float*)(ws + f.rl_box); v.rl_cls = (float*)(ws + f.rl_cls); v.G = (float*)(ws + f.G);
    v.head = (int*)(ws + f.head); v.obj_part = (float*)(ws + f.obj_part);
    v.balance = d->balance[i];
    for (int a = 0; a < d->na * 2; ++a) v.anchors[a] = d->anchors[i * Y5_LOSS_MAX_NA * 2 + a];
  }
}

}  // namespace

extern "C" size_t y5_loss_workspace_bytes(const y5_loss_desc* d, int nt) {
  if (validate(d, nt)) return 0;
  return layout(d, nt).total;
}

extern "C" long long y5_loss_obji_offset(const y5_loss_desc* d, int nt) {
  if (validate(d, nt)) return -1;
  return (long long)layout(d, nt).obji;
}

extern "C" int y5_loss_targets_layout(const y5_loss_desc* d, int nt, int level, size_t offs[10], long long* cap) {
  if (int rc = validate(d, nt)) return rc;
  if (level < 0 || level >= d->nl || !offs) return y5_fail(Y5_ERR_BAD_ARG, "loss_targets_layout: bad level");
  const Layout L = layout(d, nt);
  const LevelOff& f = L.lv[level];
  const size_t o[10] = {L.n_rows + 4 * (size_t)level, f.rb, f.ra, f.rgj, f.rgi, f.rcls, f.tbox, f.anch, f.iou, f.G};
  for (int i = 0; i < 10; ++i) offs[i] = o[i];
  if (cap) *cap = L.cap;
  return Y5_OK;
}

extern "C" int y5_loss_forward(const y5_loss_desc* d, const void* const* p, const float* targets, int nt, float* out4,
                               void* ws_, size_t ws_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = validate(d, nt)) return rc;
  if (!p || !out4 || !ws_ || (nt > 0 && !targets)) return y5_fail(Y5_ERR_BAD_ARG, "loss: null pointer");
  const Layout L = layout(d, nt);
  if (ws_bytes < L.total || ((uintptr_t)ws_ & 255)) return y5_fail(Y5_ERR_WORKSPACE, "loss: workspace too small or misaligned");
  char* ws = static_cast<char*>(ws_);
  Y5LossParams P{};
  fill(P, d, L, ws, nt);
  P.targets = targets; P.out = out4;
  for (int i = 0; i < d->nl; ++i) {
    if (!p[i]) return y5_fail(Y5_ERR_BAD_ARG, "loss: null prediction level");
    P.lv[i].p = p[i];
  }
  if (hipMemsetAsync(ws + L.head_begin, 0xFF, L.head_end - L.head_begin, st) != hipSuccess ||
      hipMemsetAsync(ws + L.n_rows, 0, sizeof(int) * Y5_LOSS_MAX_NL, st) != hipSuccess)
    return y5_fail(Y5_ERR_RUNTIME, "loss: memset failed");
  if (nt > 0) {
    hipLaunchKernelGGL(y5_loss_build_targets_kernel, dim3((unsigned)d->nl), dim3(1024), 4096, st, P);
    const unsigned rb = (unsigned)((L.cap + 3) / 4);
    for (int i = 0; i < d->nl; ++i) {
      if (d->dtype == Y5_F16) hipLaunchKernelGGL((y5_loss_rows_kernel<half_t>), dim3(rb), dim3(256), 0, st, P, i);
      else hipLaunchKernelGGL((y5_loss_rows_kernel<float>), dim3(rb), dim3(256), 0, st, P, i);
    }
  }
  for (int i = 0; i < d->nl; ++i) {
    const unsigned nb = (unsigned)((L.cells[i] + 255) / 256);
    if (d->dtype == Y5_F16) hipLaunchKernelGGL((y5_loss_obj_fwd_kernel<half_t>), dim3(nb), dim3(256), 1024, st, P, i);
    else hipLaunchKernelGGL((y5_loss_obj_fwd_kernel<float>), dim3(nb), dim3(256), 1024, st, P, i);
  }
  hipLaunchKernelGGL(y5_loss_finish_kernel, dim3(1), dim3(256), 2048, st, P);
  return y5_check_launch("y5_loss_forward");
}

extern "C" int y5_loss_backward(const y5_loss_desc* d, const void* const* p, int nt, const float* grad_scale, void* const* dp,
                                void* ws_, size_t ws_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = validate(d, nt)) return rc;
  if (!p || !dp || !ws_) return y5_fail(Y5_ERR_BAD_ARG, "loss: null pointer");
  const Layout L = layout(d, nt);
  if (ws_bytes < L.total || ((uintptr_t)ws_ & 255)) return y5_fail(Y5_ERR_WORKSPACE, "loss: workspace too small or misaligned");
  Y5LossParams P{};
  fill(P, d, L, static_cast<char*>(ws_), nt);
  P.gscale = grad_scale;
  for (int i = 0; i < d->nl; ++i) {
    if (!p[i] || !dp[i]) return y5_fail(Y5_ERR_BAD_ARG, "loss: null level pointer");
    P.lv[i].p = p[i];
    P.lv[i].dp = dp[i];
  }
  for (int i = 0; i < d->nl; ++i) {
    const unsigned nb = (unsigned)((L.cells[i] + 255) / 256);
    if (d->dtype == Y5_F16) hipLaunchKernelGGL((y5_loss_bwd_kernel<half_t>), dim3(nb), dim3(256), 2048, st, P, i);
    else hipLaunchKernelGGL((y5_loss_bwd_kernel<float>), dim3(nb), dim3(256), 2048, st, P, i);
  }
  return y5_check_launch("y5_loss_backward");
}
