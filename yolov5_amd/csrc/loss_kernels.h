// ComputeLoss (utils/loss.py:101-247) for gfx950: target assignment, CIoU / objectness / class BCE forward and the
// gradient w.r.t. the raw head outputs p[i] (bs, na, ny, nx, no), without any host synchronisation.
//
//   K1 y5_loss_build_targets_kernel   utils/loss.py:185-247.  One workgroup per level: candidate flags over
//      (offset o, anchor a, target t) in the reference's row order (offset-major, then anchor, then target),
//      block-wide exclusive scan, compacted rows (b, a, gj, gi, tcls, tbox, anch).  Integer results are bit-exact.
//   K2 y5_loss_rows_kernel            utils/loss.py:146-169 for the matched rows.  One wave per row: lane 0 does the
//      box regression + CIoU (restated ultralytics bbox_iou) and its analytic gradient, all lanes the class
//      BCE.  Rows are chained per grid cell (atomicExch list) so that duplicates can be resolved deterministically.
//   K3 y5_loss_obj_fwd_kernel         utils/loss.py:144,163,171: tobj = iou of the LAST row addressing the cell
//      (last-write-wins, SURVEY 8c hazard 3), BCE over every cell, deterministic per-block partial sums.
//   K4 y5_loss_finish_kernel          utils/loss.py:178-183: fixed-order reduction, hyp gains, x bs.
//   K5 y5_loss_bwd_kernel             d(loss * scale)/dp[i]: objectness gradient for every cell + the matched rows'
//      gradients accumulated in ascending row order (== index_put(accumulate) on the CPU), written as whole
//      contiguous rows in the dtype of p.  `scale` is read from device memory (autograd's grad_output / GradScaler).
//
// Arithmetic: fp32 on the fp32 value of every logit (fp16 inputs are widened; the reference under autocast mixes
// fp16 sigmoid with fp32 pow -- see DESIGN.md), BCE in torch's stable form (1-t)x + lw*(log1p(exp(-|x|)) + max(-x,0)).
#pragma once
#include "y5_common.h"

#define Y5_LOSS_MAX_NL 5
#define Y5_LOSS_MAX_NA 8

struct Y5LossLevel {
  const void* p;     // (bs, na, ny, nx, no)
  void* dp;          // gradient output (backward only)
  int ny, nx;
  long long cells;   // bs*na*ny*nx
  long long cap;     // row capacity = 5*na*nt
  // workspace arrays of this level
  int* rb; int* ra; int* rgj; int* rgi; int* rcls; int* next;
  float* tbox;       // [cap][4]
  float* anch;       // [cap][2]
  float* iou;        // [cap]   clamp(ciou, 0) rounded to p's dtype
  float* rl_box;     // [cap]   1 - ciou
  float* rl_cls;     // [cap]   sum_c BCE
  float* G;          // [cap][no] unit-scale gradient of the row w.r.t. its 'no' logits
  int* head;         // [cells] newest row of the cell's chain, -1 = unmatched
  float* obj_part;   // [ceil(cells/256)]
  float balance;
  float anchors[Y5_LOSS_MAX_NA * 2];  // grid units
};

struct Y5LossParams {
  Y5LossLevel lv[Y5_LOSS_MAX_NL];
  const float* targets;  // (nt, 6): img, cls, x, y, w, h (normalised)
  int* n_rows;           // [nl]
  float* out;            // [4]: loss, lbox, lobj, lcls
  float* obji;           // [nl] (workspace): each level's mean objectness BCE BEFORE its balance factor (loss.py:171) -- what autobalance reads (loss.py:173-177)
  const float* gscale;   // device scalar multiplied into the gradient (may be null = 1)
  int nl, na, nc, no, bs, nt;
  float hyp_box, hyp_obj, hyp_cls, cls_pw, obj_pw, anchor_t, cp, cn;
  float fl_gamma;  // > 0: focal loss around both BCE terms (utils/loss.py:77-98 FocalLoss(alpha = 0.25), :120-122)
};

__device__ __forceinline__ float y5_sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
// BCEWithLogits(x, t, pos_weight) element and its derivative w.r.t. x (torch: binary_cross_entropy_with_logits)
// gamma > 0: FocalLoss (utils/loss.py:77-98, alpha = 0.25 as ComputeLoss constructs it, :120-122): the element becomes
//     bce * (t a + (1-t)(1-a)) * (1 - p_t)^gamma,   p_t = t s + (1-t)(1-s),  s = sigmoid(x)
// and its derivative follows by the product rule (d p_t / dx = (2t - 1) s (1 - s)).
__device__ __forceinline__ float y5_bce(float x, float t, float pw, float& dx, float gamma = 0.0f) {
  const float lw = 1.0f + (pw - 1.0f) * t;
  const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f);  // softplus(-x)
  const float sg = y5_sigmoid_acc(x);
  const float dbce = (1.0f - t) + lw * (sg - 1.0f);
  const float bce = (1.0f - t) * x + lw * sp;
  if (!(gamma > 0.0f)) {
    dx = dbce;
    return bce;
  }
  const float af = t * 0.25f + (1.0f - t) * 0.75f;
  const float q = 1.0f - (t * sg + (1.0f - t) * (1.0f - sg));  // 1 - p_t
  const float m = powf(q, gamma);
  const float dq = -(2.0f * t - 1.0f) * sg * (1.0f - sg);
  const float dm = gamma * powf(q, gamma - 1.0f) * dq;
  dx = af * (dbce * m + bce * dm);
  return bce * af * m;
}

template <typename T> __device__ __forceinline__ float y5_round_to(float v) { return (float)(T)v; }

__device__ __forceinline__ float y5_wave_sum(float v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl(v, lane ^ m);
  return v;
}

// ---- K1 -------------------------------------------------------------------------------------------------
struct Y5Cand { bool ok; int b, a, cls, gj, gi; float tb[4]; };

__device__ __forceinline__ Y5Cand y5_loss_candidate(const Y5LossParams& p, const Y5LossLevel& L, long long c) {
  Y5Cand r;
  const int nt = p.nt;
  const int o = (int)(c / ((long long)p.na * nt));
  const int rem = (int)(c - (long long)o * p.na * nt);
  const int a = rem / nt, t = rem - a * nt;
  const float* tg = p.targets + (long long)t * 6;
  const float fnx = (float)L.nx, fny = (float)L.ny;
  const float gx = tg[2] * fnx, gy = tg[3] * fny, gw = tg[4] * fnx, gh = tg[5] * fny;  // loss.py:205
  const float rw = gw / L.anchors[a * 2], rh = gh / L.anchors[a * 2 + 1];                // loss.py:208
  const float m = fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh));
  bool ok = m < p.anchor_t;                                                              // loss.py:209
  float ox = 0.f, oy = 0.f;
  if (o == 1) { ok = ok && ((gx - floorf(gx)) < 0.5f) && (gx > 1.0f); ox = 0.5f; }       // loss.py:216 j
  else if (o == 2) { ok = ok && ((gy - floorf(gy)) < 0.5f) && (gy > 1.0f); oy = 0.5f; }  // k
  else if (o == 3) { const float gi = fnx - gx; ok = ok && ((gi - floorf(gi)) < 0.5f) && (gi > 1.0f); ox = -0.5f; }  // l
  else if (o == 4) { const float gi = fny - gy; ok = ok && ((gi - floorf(gi)) < 0.5f) && (gi > 1.0f); oy = -0.5f; }  // m
  // image index / class outside the batch / class list: the reference raises IndexError at loss.py:145,163 (device assert on a
  // GPU).  Such rows are dropped here so that no kernel of the chain can address outside p / dp (ComputeLoss.check_targets
  // raises the reference's error on the host when asked to validate).
  const int tb = (int)tg[0], tc = (int)tg[1];
  ok = ok && tb >= 0 && tb < p.bs && tc >= 0 && tc < p.nc;
  r.ok = ok;
  if (!ok) return r;
  r.b = tb;
  r.cls = tc;
  r.a = a;
  int gi = (int)(gx - ox), gj = (int)(gy - oy);                                          // loss.py:238 (.long() truncates)
  gi = gi < 0 ? 0 : gi > L.nx - 1 ? L.nx - 1 : gi;                                       // loss.py:242 clamp_ (in place: tbox
  gj = gj < 0 ? 0 : gj > L.ny - 1 ? L.ny - 1 : gj;                                       //   below sees the clamped cell)
  r.gi = gi; r.gj = gj;
  r.tb[0] = gx - (float)gi; r.tb[1] = gy - (float)gj; r.tb[2] = gw; r.tb[3] = gh;       // loss.py:243
  return r;
}

__global__ __launch_bounds__(1024)
void y5_loss_build_targets_kernel(const Y5LossParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_scan = reinterpret_cast<int*>(smem);  // [1024]
  const int lvl = blockIdx.x;
  const Y5LossLevel& L = p.lv[lvl];
  const int tid = threadIdx.x;
  const long long ncand = 5LL * p.na * p.nt;
  const long long per = (ncand + 1023) / 1024;
  const long long c0 = per * tid, c1 = c0 + per < ncand ? c0 + per : ncand;
  int cnt = 0;
  for (long long c = c0; c < c1; ++c) cnt += y5_loss_candidate(p, L, c).ok ? 1 : 0;
  s_scan[tid] = cnt;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan
    const int v = tid >= d ? s_scan[tid - d] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  int row = s_scan[tid] - cnt;
  if (tid == 1023) p.n_rows[lvl] = s_scan[1023];
  for (long long c = c0; c < c1; ++c) {
    const Y5Cand r = y5_loss_candidate(p, L, c);
    if (!r.ok) continue;
    L.rb[row] = r.b; L.ra[row] = r.a; L.rgj[row] = r.gj; L.rgi[row] = r.gi; L.rcls[row] = r.cls;
    L.tbox[row * 4 + 0] = r.tb[0]; L.tbox[row * 4 + 1] = r.tb[1]; L.tbox[row * 4 + 2] = r.tb[2]; L.tbox[row * 4 + 3] = r.tb[3];
    L.anch[row * 2 + 0] = L.anchors[r.a * 2]; L.anch[row * 2 + 1] = L.anchors[r.a * 2 + 1];
    ++row;
  }
}

// ---- K2 -------------------------------------------------------------------------------------------------
// d/dx of max(a,b) / min(a,b) routed like torch (ties split the gradient in half)
__device__ __forceinline__ void y5_gmax(float a, float b, float g, float& ga) { ga += a > b ? g : (a == b ? 0.5f * g : 0.f); }
__device__ __forceinline__ void y5_gmin(float a, float b, float g, float& ga) { ga += a < b ? g : (a == b ? 0.5f * g : 0.f); }

template <typename T>
__global__ __launch_bounds__(256)
void y5_loss_rows_kernel(const Y5LossParams p, int lvl) {
  const Y5LossLevel& L = p.lv[lvl];
  const int n = p.n_rows[lvl];
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int b = L.rb[r], a = L.ra[r], gj = L.rgj[r], gi = L.rgi[r], cls = L.rcls[r];
  const long long cell = (((long long)b * p.na + a) * L.ny + gj) * L.nx + gi;
  const T* row = static_cast<const T*>(L.p) + cell * p.no;
  float* G = L.G + (long long)r * p.no;
  const float fn = (float)n;

  // class BCE (loss.py:166-169): lanes stride the classes
  float lsum = 0.f;
  if (p.nc > 1) {
    const float gk = p.hyp_cls * (float)p.bs / (fn * (float)p.nc);
    for (int c = lane; c < p.nc; c += 64) {
      const float x = (float)row[5 + c];
      const float t = c == cls ? p.cp : p.cn;
      float dx;
      lsum += y5_bce(x, t, p.cls_pw, dx, p.fl_gamma);
      G[5 + c] = dx * gk;
    }
  } else {
    for (int c = lane; c < p.nc; c += 64) G[5 + c] = 0.f;
  }
  lsum = y5_wave_sum(lsum);

  if (lane == 0) {
    L.rl_cls[r] = lsum;
    // regression (loss.py:150-153)
    const float l0 = (float)row[0], l1 = (float)row[1], l2 = (float)row[2], l3 = (float)row[3];
    const float s0 = y5_sigmoid_acc(l0), s1 = y5_sigmoid_acc(l1), s2 = y5_sigmoid_acc(l2), s3 = y5_sigmoid_acc(l3);
    const float aw = L.anch[r * 2], ah = L.anch[r * 2 + 1];
    const float x1 = s0 * 2.0f - 0.5f, y1 = s1 * 2.0f - 0.5f;
    const float w1 = (s2 * 2.0f) * (s2 * 2.0f) * aw, h1 = (s3 * 2.0f) * (s3 * 2.0f) * ah;
    const float x2 = L.tbox[r * 4], y2 = L.tbox[r * 4 + 1], w2 = L.tbox[r * 4 + 2], h2 = L.tbox[r * 4 + 3];
    const float eps = 1e-7f;
    // bbox_iou(xywh=True, CIoU=True) -- ultralytics.utils.metrics, restated in oracle/thirdparty.py
    const float b1x1 = x1 - w1 / 2, b1x2 = x1 + w1 / 2, b1y1 = y1 - h1 / 2, b1y2 = y1 + h1 / 2;
    const float b2x1 = x2 - w2 / 2, b2x2 = x2 + w2 / 2, b2y1 = y2 - h2 / 2, b2y2 = y2 + h2 / 2;
    const float mnx2 = fminf(b1x2, b2x2), mxx1 = fmaxf(b1x1, b2x1), mny2 = fminf(b1y2, b2y2), mxy1 = fmaxf(b1y1, b2y1);
    const float iwr = mnx2 - mxx1, ihr = mny2 - mxy1;
    const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
    const float inter = iw * ih;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = inter / uni;
    const float mxx2 = fmaxf(b1x2, b2x2), mnx1 = fminf(b1x1, b2x1), mxy2 = fmaxf(b1y2, b2y2), mny1 = fminf(b1y1, b2y1);
    const float cw = mxx2 - mnx1, ch = mxy2 - mny1;
    const float c2 = cw * cw + ch * ch + eps;
    const float dxs = b2x1 + b2x2 - b1x1 - b1x2, dys = b2y1 + b2y2 - b1y1 - b1y2;
    const float rho2 = (dxs * dxs + dys * dys) / 4;
    const float kv = 0.40528473456935109f;  // 4 / pi^2
    const float q1 = w1 / h1;
    const float u = atanf(w2 / h2) - atanf(q1);
    const float v = kv * (u * u);
    const float alpha = v / (v - iou + (1.0f + eps));  // no_grad
    const float ciou = iou - (rho2 / c2 + v * alpha);
    L.rl_box[r] = 1.0f - ciou;
    L.iou[r] = y5_round_to<T>(fmaxf(ciou, 0.f));  // loss.py:157: detach().clamp(0).type(tobj.dtype)

    // reverse mode: d(hyp_box * bs * mean(1 - ciou)) / d(l0..l3)
    const float g_ciou = -p.hyp_box * (float)p.bs / fn;
    float g_iou = g_ciou;
    const float g_rho2 = -g_ciou / c2;
    const float g_c2 = g_ciou * rho2 / (c2 * c2);
    const float g_v = -g_ciou * alpha;
    float g_inter = g_iou / uni;
    const float g_uni = -g_iou * inter / (uni * uni);
    float g_w1 = g_uni * h1, g_h1 = g_uni * w1;
    g_inter -= g_uni;
    const float g_iw = iwr >= 0.f ? g_inter * ih : 0.f, g_ih = ihr >= 0.f ? g_inter * iw : 0.f;
    float g_b1x1 = 0.f, g_b1x2 = 0.f, g_b1y1 = 0.f, g_b1y2 = 0.f;
    y5_gmin(b1x2, b2x2, g_iw, g_b1x2);
    y5_gmax(b1x1, b2x1, -g_iw, g_b1x1);
    y5_gmin(b1y2, b2y2, g_ih, g_b1y2);
    y5_gmax(b1y1, b2y1, -g_ih, g_b1y1);
    const float g_cw = g_c2 * 2.0f * cw, g_ch = g_c2 * 2.0f * ch;
    y5_gmax(b1x2, b2x2, g_cw, g_b1x2);
    y5_gmin(b1x1, b2x1, -g_cw, g_b1x1);
    y5_gmax(b1y2, b2y2, g_ch, g_b1y2);
    y5_gmin(b1y1, b2y1, -g_ch, g_b1y1);
    const float g_dxs = g_rho2 * 2.0f * dxs / 4, g_dys = g_rho2 * 2.0f * dys / 4;
    g_b1x1 -= g_dxs; g_b1x2 -= g_dxs; g_b1y1 -= g_dys; g_b1y2 -= g_dys;
    const float g_u = g_v * kv * 2.0f * u;
    const float g_q1 = -g_u / (1.0f + q1 * q1);
    g_w1 += g_q1 / h1;
    g_h1 += -g_q1 * w1 / (h1 * h1);
    const float g_x1 = g_b1x1 + g_b1x2, g_y1 = g_b1y1 + g_b1y2;
    g_w1 += (g_b1x2 - g_b1x1) / 2;
    g_h1 += (g_b1y2 - g_b1y1) / 2;
    G[0] = g_x1 * 2.0f * s0 * (1.0f - s0);
    G[1] = g_y1 * 2.0f * s1 * (1.0f - s1);
    G[2] = g_w1 * aw * 8.0f * s2 * s2 * (1.0f - s2);
    G[3] = g_h1 * ah * 8.0f * s3 * s3 * (1.0f - s3);
    G[4] = 0.f;
    L.next[r] = atomicExch(L.head + cell, r);
  }
}

// newest-first chain walk: the LAST build_targets row addressing the cell (largest row id) defines tobj
__device__ __forceinline__ int y5_loss_winner(const Y5LossLevel& L, int h) {
  int w = -1;
  while (h >= 0) { w = h > w ? h : w; h = L.next[h]; }
  return w;
}

// ---- K3 -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void y5_loss_obj_fwd_kernel(const Y5LossParams p, int lvl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_red = reinterpret_cast<float*>(smem);  // [256]
  const Y5LossLevel& L = p.lv[lvl];
  const long long cell = (long long)blockIdx.x * 256 + threadIdx.x;
  float l = 0.f;
  if (cell < L.cells) {
    const float x = (float)static_cast<const T*>(L.p)[cell * p.no + 4];
    const int w = y5_loss_winner(L, L.head[cell]);
    const float t = w >= 0 ? L.iou[w] : 0.f;
    float dx;
    l = y5_bce(x, t, p.obj_pw, dx, p.fl_gamma);
  }
  s_red[threadIdx.x] = l;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) L.obj_part[blockIdx.x] = s_red[0];
}

// ---- K4 -------------------------------------------------------------------------------------------------
__device__ __forceinline__ double y5_block_sum_f(const float* v, long long n, double* s_red) {
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) s += (double)v[i];
  s_red[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
    __syncthreads();
  }
  const double r = s_red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256)
void y5_loss_finish_kernel(const Y5LossParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* s_red = reinterpret_cast<double*>(smem);  // [256]
  float lbox = 0.f, lobj = 0.f, lcls = 0.f;
  for (int i = 0; i < p.nl; ++i) {
    const Y5LossLevel& L = p.lv[i];
    const int n = p.n_rows[i];
    if (n > 0) {
      lbox += (float)(y5_block_sum_f(L.rl_box, n, s_red) / (double)n);                       // loss.py:154
      if (p.nc > 1) lcls += (float)(y5_block_sum_f(L.rl_cls, n, s_red) / ((double)n * p.nc));  // loss.py:169
    }
    const long long nblk = (L.cells + 255) / 256;
    const float obji = (float)(y5_block_sum_f(L.obj_part, nblk, s_red) / (double)L.cells);   // loss.py:171
    lobj += obji * L.balance;                                                                // loss.py:172
    if (threadIdx.x == 0) p.obji[i] = obji;
  }
  if (threadIdx.x == 0) {
    lbox *= p.hyp_box; lobj *= p.hyp_obj; lcls *= p.hyp_cls;                                 // loss.py:178-180
    p.out[0] = (lbox + lobj + lcls) * (float)p.bs;                                           // loss.py:183
    p.out[1] = lbox; p.out[2] = lobj; p.out[3] = lcls;
  }
}

// ---- K5 -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void y5_loss_bwd_kernel(const Y5LossParams p, int lvl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_g = reinterpret_cast<float*>(smem);        // [256]
  int* s_head = reinterpret_cast<int*>(smem + 1024);  // [256]
  const Y5LossLevel& L = p.lv[lvl];
  const int tid = threadIdx.x;
  const long long cell0 = (long long)blockIdx.x * 256;
  const long long cell = cell0 + tid;
  const int ncell = L.cells - cell0 < 256 ? (int)(L.cells - cell0) : 256;
  const float scale = p.gscale ? *p.gscale : 1.0f;
  const float gobj = p.hyp_obj * L.balance * (float)p.bs / (float)L.cells;
  const T* P = static_cast<const T*>(L.p);
  T* D = static_cast<T*>(L.dp);
  int h = -1;
  float g4 = 0.f;
  if (tid < ncell) {
    const float x = (float)P[cell * p.no + 4];
    h = L.head[cell];
    const int w = y5_loss_winner(L, h);
    const float t = w >= 0 ? L.iou[w] : 0.f;
    float dx;
    y5_bce(x, t, p.obj_pw, dx, p.fl_gamma);
    g4 = dx * gobj;
  }
  s_g[tid] = g4;
  s_head[tid] = h;
  __syncthreads();
  // unmatched cells: a contiguous run of ncell*no elements, zero except the objectness slot
  const int ne = ncell * p.no;
  T* Dblk = D + cell0 * p.no;
  for (int e = tid; e < ne; e += 256) {
    const int cl = e / p.no, o = e - cl * p.no;
    if (s_head[cl] >= 0) continue;
    Dblk[e] = (T)(o == 4 ? s_g[cl] * scale : 0.f);
  }
  // matched cells: sum the chain's rows in ascending row order
  if (h >= 0) {
    int rows[16];
    int k = 0;
    for (int q = h; q >= 0; q = L.next[q]) { if (k < 16) rows[k] = q; ++k; }
    T* d = D + cell * p.no;
    if (k <= 16) {
      for (int i = 1; i < k; ++i) {  // insertion sort ascending
        const int v = rows[i];
        int j = i - 1;
        while (j >= 0 && rows[j] > v) { rows[j + 1] = rows[j]; --j; }
        rows[j + 1] = v;
      }
      for (int o = 0; o < p.no; ++o) {
        float s = 0.f;
        for (int i = 0; i < k; ++i) s += L.G[(long long)rows[i] * p.no + o];
        if (o == 4) s += g4;
        d[o] = (T)(s * scale);
      }
    } else {  // very long chain: repeated selection of the next larger row id
      for (int o = 0; o < p.no; ++o) {
        float s = 0.f;
        int last = -1;
        for (int i = 0; i < k; ++i) {
          int best = 0x7fffffff;
          for (int q = h; q >= 0; q = L.next[q]) if (q > last && q < best) best = q;
          s += L.G[(long long)best * p.no + o];
          last = best;
        }
        if (o == 4) s += g4;
        d[o] = (T)(s * scale);
      }
    }
  }
}
