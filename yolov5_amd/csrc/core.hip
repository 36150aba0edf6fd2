// Library core: status/error reporting, the per-device zero page, and the execution plan
// (recorded op list replayed by one host call, optionally through a captured hipGraph).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/yolov5_hip.h"
#include "y5_host.h"

namespace {
thread_local std::string g_err;
std::mutex g_zero_mu;
void* g_zero[64] = {nullptr};
}  // namespace

int y5_fail(int code, const char* msg) {
  g_err = msg ? msg : "unknown error";
  return code;
}

int y5_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return Y5_OK;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return Y5_ERR_RUNTIME;
}

const void* y5_zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(g_zero_mu);
  if (!g_zero[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess) return nullptr;
    g_zero[dev] = p;
  }
  return g_zero[dev];
}

extern "C" int y5_version(void) { return 10000 * 0 + 100 * 1 + 0; }
extern "C" const char* y5_last_error(void) { return g_err.c_str(); }

// ---- CU budget: persistent kernels size their grids as (CUs x occupancy).  Under data-parallel training the gradient all-reduce (RCCL's own kernels,
// on the process group's stream) runs BESIDE the backward plan; with every CU held by a persistent workgroup for the whole life of each launch the
// collective's workgroups only get in at kernel boundaries (measured through a one-rank group: 28.9 MB exposed for 0.73 ms, profiles/r04).  A budget
// of CUs - r leaves r CUs' worth of workgroup slots to the collective (utils/torch_utils.py:61-70 smart_DDP; yolov5_amd.torch_utils.HipDDP sets it).
namespace { int g_cu_budget = 0; }
extern "C" int y5_set_cu_budget(int n_cus) {
  if (n_cus < 0) return y5_fail(Y5_ERR_BAD_ARG, "set_cu_budget: negative");
  g_cu_budget = n_cus;   // 0 = every CU of the device
  return Y5_OK;
}
int y5_num_cu() {
  static int cu_of_dev[64] = {};   // per device (the count of the first device queried was served for all)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cu_of_dev[dev]) {
    int n = 0;
    hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cu_of_dev[dev] = n > 0 ? n : 256;
  }
  const int n = cu_of_dev[dev];
  return g_cu_budget > 0 && g_cu_budget < n ? g_cu_budget : n;
}

// ---------------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------------
enum OpKind { OP_CONV, OP_TO_NHWC, OP_SPPF, OP_UPS, OP_COPY, OP_DECODE, OP_TO_NCHW, OP_STEM, OP_HEAD, OP_NOP, OP_BNECK, OP_K3PW, OP_BNECK_CV3, OP_FRONT, OP_SPPF_FRONT };

struct Op {
  OpKind kind;
  y5_conv_desc conv;
  const void* p0; const void* p1; const void* p2; const void* p3; void* q0; void* q1;
  const void* r0; const void* r1; const void* r2;  // further read-only operands (OP_BNECK_CV3: y2, w3, bias3)
  int i[16];
  float f[2];
  long long l[2];
  float anchors[16];
  int branch;  // 0: the caller's stream; 1: the plan's side stream (forked after the preceding main op, joined at the end of the range)
};

struct GraphEntry { unsigned long long key; hipGraph_t graph; hipGraphExec_t exec; unsigned long long used; };

struct y5_plan {
  std::vector<Op> ops;
  hipGraph_t graph = nullptr;        // the SELECTED graph (one entry of `cache`, or the only graph of a plan that never rebinds)
  hipGraphExec_t exec = nullptr;
  std::vector<GraphEntry> cache;     // captured graphs by binding key (y5_plan_select_graph): outputs re-pointed per call keep their graphs
  unsigned long long cur_key = 0, tick = 0;
  hipStream_t side = nullptr;        // side-branch stream (y5_plan_set_branch), created on first use
  std::vector<hipEvent_t> events;    // fork / join markers, one per fork point of a run + one join
  bool flat = false;                 // run every op on the caller's stream (per-op timing: no fork / join latency in the figure)
};

extern "C" y5_plan* y5_plan_create(void) { return new y5_plan(); }
extern "C" void y5_plan_destroy(y5_plan* p) {
  if (!p) return;
  for (GraphEntry& g : p->cache) {  // (p->graph / p->exec alias one entry)
    if (g.exec) hipGraphExecDestroy(g.exec);
    if (g.graph) hipGraphDestroy(g.graph);
  }
  for (hipEvent_t e : p->events) hipEventDestroy(e);
  if (p->side) hipStreamDestroy(p->side);
  delete p;
}
extern "C" int y5_plan_size(const y5_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int y5_plan_add_conv(y5_plan* pl, const y5_conv_desc* d, const void* x, const void* w, const float* bias,
                                const void* res, void* y, void* y2) {
  if (!pl || !d) return y5_fail(Y5_ERR_BAD_ARG, "plan_add_conv: null");
  Op o{}; o.kind = OP_CONV; o.conv = *d; o.p0 = x; o.p1 = w; o.p2 = bias; o.p3 = res; o.q0 = y; o.q1 = y2;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_nchw_to_nhwc(y5_plan* pl, const void* src, int sdt, void* dst, int ddt, int B, int C, int H, int W,
                                        int ld, float scale) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_TO_NHWC; o.p0 = src; o.q0 = dst;
  o.i[0] = sdt; o.i[1] = ddt; o.i[2] = B; o.i[3] = C; o.i[4] = H; o.i[5] = W; o.i[6] = ld; o.f[0] = scale;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_nhwc_to_nchw(y5_plan* pl, const void* src, int dt, void* dst, int B, int C, int H, int W, int ld) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_TO_NCHW; o.p0 = src; o.q0 = dst;
  o.i[0] = dt; o.i[1] = B; o.i[2] = C; o.i[3] = H; o.i[4] = W; o.i[5] = ld;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_sppf_cv1_pool(y5_plan* pl, const void* x, int ldx, const void* w, const float* bias, int Kpad, void* buf, int ld, int B, int H, int W,
                                         int C1, int c_, int k, int act) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_SPPF_FRONT; o.p0 = x; o.p1 = w; o.p2 = bias; o.q0 = buf;
  o.i[0] = ldx; o.i[1] = Kpad; o.i[2] = ld; o.i[3] = B; o.i[4] = H; o.i[5] = W; o.i[6] = C1; o.i[7] = c_; o.i[8] = k; o.i[9] = act;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_sppf_pool(y5_plan* pl, void* buf, int dt, int B, int H, int W, int C, int ld, int k) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_SPPF; o.q0 = buf;
  o.i[0] = dt; o.i[1] = B; o.i[2] = H; o.i[3] = W; o.i[4] = C; o.i[5] = ld; o.i[6] = k;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_upsample2x(y5_plan* pl, const void* src, int dt, void* dst, int B, int H, int W, int C, int lds, int ldd) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_UPS; o.p0 = src; o.q0 = dst;
  o.i[0] = dt; o.i[1] = B; o.i[2] = H; o.i[3] = W; o.i[4] = C; o.i[5] = lds; o.i[6] = ldd;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_copy_slice(y5_plan* pl, const void* src, int dt, void* dst, int npix, int C, int lds, int ldd) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_COPY; o.p0 = src; o.q0 = dst;
  o.i[0] = dt; o.i[1] = npix; o.i[2] = C; o.i[3] = lds; o.i[4] = ldd;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_detect_decode(y5_plan* pl, const void* logits, int dt, int B, int ny, int nx, int na, int no, int nm, int ld,
                                         float stride, const float* anchors_px, void* z, int zdt, long long nrows_total,
                                         long long row_off, void* raw) {
  if (!pl || !anchors_px || na > 8) return y5_fail(Y5_ERR_BAD_ARG, "plan: null / na");
  Op o{}; o.kind = OP_DECODE; o.p0 = logits; o.q0 = z; o.q1 = raw;
  o.i[0] = dt; o.i[1] = B; o.i[2] = ny; o.i[3] = nx; o.i[4] = na; o.i[5] = no; o.i[6] = nm; o.i[7] = ld; o.i[8] = zdt;
  o.f[0] = stride; o.l[0] = nrows_total; o.l[1] = row_off;
  for (int k = 0; k < na * 2; ++k) o.anchors[k] = anchors_px[k];
  pl->ops.push_back(o);
  return Y5_OK;
}

// fused Detect head (head.hip): the convolution `d` + the decode of one level; y5_plan_add_nop keeps the op numbering of the
// two-op form (convolution, decode) so that a plan built either way has the same indices
extern "C" int y5_plan_add_detect_head(y5_plan* pl, const y5_conv_desc* d, const void* x, const void* w, const float* bias, int ny, int nx,
                                       float stride, const float* anchors_px, void* z, long long nrows_total, long long row_off) {
  if (!pl || !d || !anchors_px) return y5_fail(Y5_ERR_BAD_ARG, "plan_add_detect_head: null");
  Op o{}; o.kind = OP_HEAD; o.conv = *d; o.p0 = x; o.p1 = w; o.p2 = bias; o.q0 = z;
  o.i[0] = ny; o.i[1] = nx; o.f[0] = stride; o.l[0] = nrows_total; o.l[1] = row_off;
  for (int k = 0; k < 6; ++k) o.anchors[k] = anchors_px[k];
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_bottleneck(y5_plan* pl, const void* x, int ldx, const void* w1, const float* b1, int Kpad1, const void* w2, const float* b2,
                                      int Kpad2, void* y, int ldy, int B, int H, int W, int C, int add) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_BNECK; o.p0 = x; o.p1 = w1; o.p2 = b1; o.p3 = w2; o.q0 = y; o.q1 = const_cast<float*>(b2);
  o.i[0] = ldx; o.i[1] = Kpad1; o.i[2] = Kpad2; o.i[3] = ldy; o.i[4] = B; o.i[5] = H; o.i[6] = W; o.i[7] = C; o.i[8] = add;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_conv_k3pw(y5_plan* pl, const y5_conv_desc* d, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                     int C3, int Npad2, int Kpad2, int act2, void* y, int ldy, void* y2, int ld2, int split_n) {
  if (!pl || !d) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_K3PW; o.conv = *d; o.p0 = x; o.p1 = w1; o.p2 = b1; o.p3 = w2; o.q0 = y; o.q1 = y2;
  o.l[0] = (long long)(uintptr_t)b2;
  o.i[0] = C3; o.i[1] = Npad2; o.i[2] = Kpad2; o.i[3] = act2; o.i[4] = ldy; o.i[5] = ld2; o.i[6] = split_n;
  pl->ops.push_back(o);
  return Y5_OK;
}
// Detect decode / fused head op `op`: also write every row's objectness to `hint` ((B, nrows_total), z's dtype) -- the NMS filter's shortcut
extern "C" int y5_plan_set_obj_hint(y5_plan* pl, int op, void* hint) {
  if (!pl || op < 0 || op >= (int)pl->ops.size() || (pl->ops[op].kind != OP_DECODE && pl->ops[op].kind != OP_HEAD))
    return y5_fail(Y5_ERR_BAD_ARG, "plan_set_obj_hint: not a Detect decode / fused head op");
  pl->ops[op].p3 = hint;
  return Y5_OK;
}
extern "C" int y5_plan_add_bottleneck_cv3(y5_plan* pl, const void* x, int ldx, const void* w1, const float* b1, int Kpad1, const void* w2, const float* b2,
                                          int Kpad2, const void* y2, int ld2, const void* w3, const float* b3, int Kpad3, int C3, int act3, void* out, int ldo,
                                          int B, int H, int W, int C, int add) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_BNECK_CV3; o.p0 = x; o.p1 = w1; o.p2 = b1; o.p3 = w2; o.q0 = out; o.q1 = const_cast<float*>(b2);
  o.r0 = y2; o.r1 = w3; o.r2 = b3;
  o.i[0] = ldx; o.i[1] = Kpad1; o.i[2] = Kpad2; o.i[3] = ldo; o.i[4] = B; o.i[5] = H; o.i[6] = W; o.i[7] = C; o.i[8] = add;
  o.i[9] = ld2; o.i[10] = Kpad3; o.i[11] = C3 | (act3 ? 1 << 16 : 0);
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_conv_front(y5_plan* pl, const void* x_nchw, int B, int H, int W, const void* w_stem, const float* bias0, int C0, const void* w1,
                                      const float* bias1, int C1, int Npad1, int Kpad1, int act1, const void* w2, const float* bias2, int C3, int Npad2,
                                      int Kpad2, int act2, void* y, int ldy, void* y2, int ld2, int split_n) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_FRONT; o.p0 = x_nchw; o.p1 = w_stem; o.p2 = bias0; o.p3 = w1; o.r0 = bias1; o.r1 = w2; o.r2 = bias2; o.q0 = y; o.q1 = y2;
  o.i[0] = B; o.i[1] = H; o.i[2] = W; o.i[3] = C0; o.i[4] = C1; o.i[5] = Npad1; o.i[6] = Kpad1; o.i[7] = act1; o.i[8] = C3; o.i[9] = Npad2; o.i[10] = Kpad2;
  o.i[11] = act2; o.i[12] = ldy; o.i[13] = ld2; o.i[14] = split_n;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_add_nop(y5_plan* pl) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_NOP;
  pl->ops.push_back(o);
  return Y5_OK;
}

extern "C" int y5_plan_add_conv_stem(y5_plan* pl, const void* x_nchw, int B, int H, int W, const void* w_stem, const float* bias,
                                     int C2, int Npad, void* y, int ldy) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan: null");
  Op o{}; o.kind = OP_STEM; o.p0 = x_nchw; o.p1 = w_stem; o.p2 = bias; o.q0 = y;
  o.i[0] = B; o.i[1] = H; o.i[2] = W; o.i[3] = C2; o.i[4] = Npad; o.i[5] = ldy;
  pl->ops.push_back(o);
  return Y5_OK;
}
extern "C" int y5_plan_set_input(y5_plan* pl, int op, const void* src) {
  if (!pl || op < 0 || op >= (int)pl->ops.size()) return y5_fail(Y5_ERR_BAD_ARG, "plan_set_input: bad op index");
  if (pl->ops[op].kind != OP_STEM && pl->ops[op].kind != OP_TO_NHWC && pl->ops[op].kind != OP_FRONT) return y5_fail(Y5_ERR_BAD_ARG, "plan_set_input: op does not read the model input");
  pl->ops[op].p0 = src;
  return Y5_OK;
}

// Re-point every output pointer of ops [first, last) that equals `old_ptr` at `new_ptr` (the Detect decode / fused head / to_nchw
// destinations: the model returns a FRESH z / proto tensor per call like the reference, models/yolo.py:115, without a copy).
extern "C" int y5_plan_rebind_output(y5_plan* pl, int first, int last, const void* old_ptr, void* new_ptr) {
  if (!pl || first < 0 || last > (int)pl->ops.size() || first > last || !old_ptr || !new_ptr) return y5_fail(Y5_ERR_BAD_ARG, "plan_rebind_output: bad args");
  int n = 0;
  for (int k = first; k < last; ++k) {
    Op& o = pl->ops[k];
    if (o.q0 == old_ptr) { o.q0 = new_ptr; ++n; }
    if (o.q1 == old_ptr && o.kind != OP_BNECK && o.kind != OP_BNECK_CV3) { o.q1 = new_ptr; ++n; }
    if (o.p3 == old_ptr && (o.kind == OP_DECODE || o.kind == OP_HEAD)) { o.p3 = new_ptr; ++n; }  // objectness hint plane (y5_plan_set_obj_hint)
  }
  if (!n) return y5_fail(Y5_ERR_BAD_ARG, "plan_rebind_output: no op writes that pointer");
  return Y5_OK;
}

// Graph cache: select the captured graph recorded under `key` (the caller's hash of the current output bindings).  Returns 1 when
// such a graph exists (y5_plan_launch_graph replays it), 0 when not (capture again: it is stored under `key`), < 0 on error.
// At most Y5_GRAPH_CACHE graphs are kept per plan; the least recently selected one is dropped.
#define Y5_GRAPH_CACHE 8
extern "C" int y5_plan_select_graph(y5_plan* pl, unsigned long long key) {
  if (!pl) return y5_fail(Y5_ERR_BAD_ARG, "plan_select_graph: null");
  pl->cur_key = key;
  for (GraphEntry& g : pl->cache)
    if (g.key == key) {
      g.used = ++pl->tick;
      pl->graph = g.graph; pl->exec = g.exec;
      return 1;
    }
  pl->graph = nullptr; pl->exec = nullptr;
  return 0;
}

// New anchor sizes (pixels) for a Detect decode / fused head op: Detect.anchors is a buffer the EMA interpolates like any other
// (utils/torch_utils.py:361-365), so a refreshed plan must pick it up together with the filters.
extern "C" int y5_plan_set_anchors(y5_plan* pl, int op, const float* anchors_px, int n) {
  if (!pl || op < 0 || op >= (int)pl->ops.size() || !anchors_px || n < 1 || n > 16) return y5_fail(Y5_ERR_BAD_ARG, "plan_set_anchors: bad args");
  Op& o = pl->ops[op];
  if (o.kind != OP_DECODE && o.kind != OP_HEAD) return y5_fail(Y5_ERR_BAD_ARG, "plan_set_anchors: op has no anchors");
  for (int k = 0; k < n; ++k) o.anchors[k] = anchors_px[k];
  return Y5_OK;
}

static int run_op(const Op& o, void* st) {
  switch (o.kind) {
    case OP_STEM: return y5_conv_stem_fwd(o.p0, o.i[0], o.i[1], o.i[2], o.p1, (const float*)o.p2, o.i[3], o.i[4], o.q0, o.i[5], 0, st);
    case OP_CONV: return y5_conv2d_fwd(&o.conv, o.p0, o.p1, (const float*)o.p2, o.p3, o.q0, o.q1, st);
    case OP_TO_NHWC: return y5_nchw_to_nhwc(o.p0, o.i[0], o.q0, o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.f[0], st);
    case OP_TO_NCHW: return y5_nhwc_to_nchw(o.p0, o.i[0], o.q0, o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], st);
    case OP_SPPF: return y5_sppf_pool(o.q0, o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], st);
    case OP_UPS: return y5_upsample2x(o.p0, o.i[0], o.q0, o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], st);
    case OP_COPY: return y5_copy_slice(o.p0, o.i[0], o.q0, o.i[1], o.i[2], o.i[3], o.i[4], st);
    case OP_HEAD:
      return y5_detect_head_fwd_hint(&o.conv, o.p0, o.p1, (const float*)o.p2, o.i[0], o.i[1], o.f[0], o.anchors, o.q0, o.l[0], o.l[1], const_cast<void*>(o.p3), st);
    case OP_NOP: return Y5_OK;
    case OP_SPPF_FRONT:
      return y5_sppf_cv1_pool_fwd(o.p0, o.i[0], o.p1, (const float*)o.p2, o.i[1], o.q0, o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.i[7], o.i[8], o.i[9], st);
    case OP_BNECK_CV3:
      return y5_bottleneck_cv3_fwd(o.p0, o.i[0], o.p1, (const float*)o.p2, o.i[1], o.p3, (const float*)o.q1, o.i[2], o.r0, o.i[9], o.r1, (const float*)o.r2,
                                   o.i[10], o.i[11] & 0xffff, o.i[11] >> 16, o.q0, o.i[3], o.i[4], o.i[5], o.i[6], o.i[7], o.i[8], 0, st);
    case OP_FRONT:
      return y5_conv_front_fwd(o.p0, o.i[0], o.i[1], o.i[2], o.p1, (const float*)o.p2, o.i[3], o.p3, (const float*)o.r0, o.i[4], o.i[5], o.i[6], o.i[7], o.r1,
                               (const float*)o.r2, o.i[8], o.i[9], o.i[10], o.i[11], o.q0, o.i[12], o.q1, o.i[13], o.i[14], 0, st);
    case OP_K3PW:
      return y5_conv_k3pw_fwd(&o.conv, o.p0, o.p1, (const float*)o.p2, o.p3, (const float*)(uintptr_t)o.l[0], o.i[0], o.i[1], o.i[2], o.i[3], o.q0, o.i[4],
                              o.q1, o.i[5], o.i[6], st);
    case OP_BNECK:
      return y5_bottleneck_fwd(o.p0, o.i[0], o.p1, (const float*)o.p2, o.i[1], o.p3, (const float*)o.q1, o.i[2], o.q0, o.i[3], o.i[4], o.i[5], o.i[6],
                               o.i[7], o.i[8], 0, st);
    case OP_DECODE:
      return y5_detect_decode_hint(o.p0, o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.i[7], o.f[0], o.anchors, o.q0, o.i[8],
                                   o.l[0], o.l[1], o.q1, const_cast<void*>(o.p3), st);
  }
  return y5_fail(Y5_ERR_BAD_ARG, "plan: unknown op");
}

extern "C" int y5_plan_set_branch(y5_plan* pl, int op, int branch) {
  if (!pl || op < 0 || op >= (int)pl->ops.size() || branch < 0 || branch > 1) return y5_fail(Y5_ERR_BAD_ARG, "plan_set_branch: bad op index / branch");
  pl->ops[op].branch = branch;
  return Y5_OK;
}

extern "C" int y5_plan_set_conv_cfg(y5_plan* pl, int op, int cfg) {
  if (!pl || op < 0 || op >= (int)pl->ops.size() || pl->ops[op].kind != OP_CONV) return y5_fail(Y5_ERR_BAD_ARG, "plan_set_conv_cfg: not a convolution op");
  if (pl->graph || !pl->cache.empty()) return y5_fail(Y5_ERR_BAD_ARG, "plan_set_conv_cfg: the plan has captured graphs");
  pl->ops[op].conv.cfg = cfg;
  return Y5_OK;
}

static hipEvent_t plan_event(y5_plan* pl, size_t idx) {
  while (pl->events.size() <= idx) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    pl->events.push_back(e);
  }
  return pl->events[idx];
}

// Ops run in list order on `st`; a run of consecutive branch-1 ops is forked onto the plan's side stream behind the main op that
// precedes it (event record / wait) and main continues with the ops after the run; the side stream is joined back into `st` at
// the end of the range.  The host (engine.py) marks only ops whose inputs are complete at the fork point and whose outputs no
// later op of the range reads.  Under stream capture the same calls become the fork / join edges of the graph.
extern "C" int y5_plan_run_range(y5_plan* pl, int first, int last, void* st_) {
  if (!pl || first < 0 || last > (int)pl->ops.size() || first > last) return y5_fail(Y5_ERR_BAD_ARG, "plan_run_range: bad range");
  hipStream_t st = static_cast<hipStream_t>(st_);
  size_t nev = 0;
  bool used_side = false;
  for (int k = first; k < last; ++k) {
    const Op& o = pl->ops[k];
    if (o.branch == 1 && !pl->flat) {
      if (!pl->side && hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "plan: side stream create failed");
      if (k == first || pl->ops[k - 1].branch == 0) {
        hipEvent_t e = plan_event(pl, nev++);
        if (!e || hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(pl->side, e, 0) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "plan: fork failed");
      }
      used_side = true;
      const int rc = run_op(o, pl->side);
      if (rc) return rc;
    } else {
      const int rc = run_op(o, st_);
      if (rc) return rc;
    }
  }
  if (used_side) {
    hipEvent_t e = plan_event(pl, nev++);
    if (!e || hipEventRecord(e, pl->side) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "plan: join failed");
  }
  return Y5_OK;
}
extern "C" int y5_plan_run(y5_plan* pl, void* st) { return y5_plan_run_range(pl, 0, pl ? (int)pl->ops.size() : 0, st); }

extern "C" int y5_plan_capture_range(y5_plan* pl, int first, int last, void* st_) {
  if (!pl || first < 0 || last > (int)pl->ops.size() || first >= last) return y5_fail(Y5_ERR_BAD_ARG, "plan_capture: bad range");
  hipStream_t st = static_cast<hipStream_t>(st_);
  if (!y5_zero_page()) return y5_fail(Y5_ERR_RUNTIME, "plan_capture: zero page");
  // one eager run first: performs the per-kernel one-time attribute setup outside of capture
  int rc = y5_plan_run_range(pl, first, last, st_);
  if (rc) return rc;
  if (hipStreamSynchronize(st) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "plan_capture: sync failed");
  for (size_t k = 0; k < pl->cache.size(); ++k)
    if (pl->cache[k].key == pl->cur_key) {  // re-capture under the same key: drop the stale entry
      if (pl->cache[k].exec) hipGraphExecDestroy(pl->cache[k].exec);
      if (pl->cache[k].graph) hipGraphDestroy(pl->cache[k].graph);
      pl->cache.erase(pl->cache.begin() + k);
      break;
    }
  pl->exec = nullptr; pl->graph = nullptr;  // aliases of cache entries
  // capture on a private stream (the caller's may be the legacy default stream, which cannot be captured); the graph itself
  // is stream-agnostic and is launched on whatever stream y5_plan_launch_graph receives
  hipStream_t cs = nullptr;
  if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "plan_capture: stream create failed");
  if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    hipStreamDestroy(cs);
    return y5_fail(Y5_ERR_RUNTIME, "plan_capture: begin capture failed");
  }
  rc = y5_plan_run_range(pl, first, last, cs);
  const hipError_t e = hipStreamEndCapture(cs, &pl->graph);
  hipStreamDestroy(cs);
  if (rc) return rc;
  if (e != hipSuccess || !pl->graph) {
    pl->graph = nullptr;
    return y5_fail(Y5_ERR_RUNTIME, "plan_capture: end capture failed");
  }
  if (hipGraphInstantiate(&pl->exec, pl->graph, nullptr, nullptr, 0) != hipSuccess) {
    hipGraphDestroy(pl->graph);
    pl->graph = nullptr; pl->exec = nullptr;
    return y5_fail(Y5_ERR_RUNTIME, "plan_capture: instantiate failed");
  }
  if (pl->cache.size() >= Y5_GRAPH_CACHE) {
    size_t lru = 0;
    for (size_t k = 1; k < pl->cache.size(); ++k)
      if (pl->cache[k].used < pl->cache[lru].used) lru = k;
    if (pl->cache[lru].exec) hipGraphExecDestroy(pl->cache[lru].exec);
    if (pl->cache[lru].graph) hipGraphDestroy(pl->cache[lru].graph);
    pl->cache.erase(pl->cache.begin() + lru);
  }
  pl->cache.push_back({pl->cur_key, pl->graph, pl->exec, ++pl->tick});
  return Y5_OK;
}
extern "C" int y5_plan_capture(y5_plan* pl, void* st_) { return y5_plan_capture_range(pl, 0, pl ? (int)pl->ops.size() : 0, st_); }

extern "C" int y5_plan_launch_graph(y5_plan* pl, void* st_) {
  if (!pl || !pl->exec) return y5_fail(Y5_ERR_BAD_ARG, "plan_launch_graph: plan not captured");
  if (hipGraphLaunch(pl->exec, static_cast<hipStream_t>(st_)) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "plan_launch_graph: launch failed");
  return Y5_OK;
}

extern "C" int y5_plan_time_range(y5_plan* pl, int first, int last, int iters, void* st_, float* ms) {
  if (!pl || !ms || iters < 1) return y5_fail(Y5_ERR_BAD_ARG, "plan_time_range: bad args");
  hipStream_t st = static_cast<hipStream_t>(st_);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "event create failed");
  pl->flat = true;  // kernels timed on `st` itself
  int rc = y5_plan_run_range(pl, first, last, st_);  // warm-up
  if (!rc) {
    hipEventRecord(e0, st);
    for (int it = 0; it < iters && !rc; ++it) rc = y5_plan_run_range(pl, first, last, st_);
    hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) rc = y5_fail(Y5_ERR_RUNTIME, "event sync failed");
    else hipEventElapsedTime(ms, e0, e1);
  }
  pl->flat = false;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}

// In-situ per-op timing: ONE eager pass over ops [first, last) on `st` with a HIP event between consecutive ops, repeated `iters`
// times; ms_out[k - first] = median over the passes of op k's event-to-event time.  Unlike y5_plan_time_range (back-to-back
// launches of one op on warm buffers) every op here runs in its real position: inputs produced by the previous op, caches in the
// state the previous layer left them.  Side-branch ops run on `st` too (serialised), so the figures add up to a single-stream
// forward.  The cost of the event record itself (an empty event-to-event interval, measured in the same passes) is subtracted.
// Synchronises the stream.
extern "C" int y5_plan_profile_range(y5_plan* pl, int first, int last, int iters, void* st_, float* ms_out) {
  if (!pl || !ms_out || iters < 1 || first < 0 || last > (int)pl->ops.size() || first >= last) return y5_fail(Y5_ERR_BAD_ARG, "plan_profile_range: bad args");
  hipStream_t st = static_cast<hipStream_t>(st_);
  const int n = last - first;
  std::vector<hipEvent_t> ev(n + 2, nullptr);  // ev[n] -> ev[n + 1]: nothing in between = the cost of the event pair itself
  for (int k = 0; k <= n + 1; ++k)
    if (hipEventCreate(&ev[k]) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "plan_profile_range: event create failed");
  std::vector<std::vector<float>> samples(n);
  std::vector<float> empty;
  int rc = Y5_OK;
  pl->flat = true;
  rc = y5_plan_run_range(pl, first, last, st_);  // warm-up pass
  for (int it = 0; it < iters && !rc; ++it) {
    hipEventRecord(ev[0], st);
    for (int k = 0; k < n && !rc; ++k) {
      rc = run_op(pl->ops[first + k], st_);
      hipEventRecord(ev[k + 1], st);
    }
    hipEventRecord(ev[n + 1], st);
    if (!rc && hipEventSynchronize(ev[n + 1]) != hipSuccess) rc = y5_fail(Y5_ERR_RUNTIME, "plan_profile_range: sync failed");
    for (int k = 0; k < n && !rc; ++k) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
      samples[k].push_back(ms);
    }
    if (!rc) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, ev[n], ev[n + 1]);
      empty.push_back(ms);
    }
  }
  pl->flat = false;
  auto median = [](std::vector<float>& v) {
    for (size_t a = 1; a < v.size(); ++a)  // insertion sort (iters is small)
      for (size_t b = a; b > 0 && v[b] < v[b - 1]; --b) { const float t = v[b]; v[b] = v[b - 1]; v[b - 1] = t; }
    return v.empty() ? 0.f : v[v.size() / 2];
  };
  // an interval = [dispatch + run of the op] + [one event record]; the record's own cost (measured on the empty interval, ~5 us:
  // the completion signal + timestamp write-back of the marker packet) is taken off every op
  const float ev_cost = median(empty);
  for (int k = 0; k < n && !rc; ++k) {
    const float m = median(samples[k]) - ev_cost;
    ms_out[k] = m > 0.f ? m : 0.f;
  }
  for (hipEvent_t e : ev) hipEventDestroy(e);
  return rc;
}
