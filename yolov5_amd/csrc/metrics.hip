// Validation matching on the device: val.py:296-307 per image
//     predn = pred.clone(); scale_boxes(im.shape, predn[:, :4], shape0, ratio_pad)          native-space predictions
//     tbox  = xywh2xyxy(labels[:, 1:5]); scale_boxes(im.shape, tbox, shape0, ratio_pad)     native-space labels
//     correct = process_batch(predn, labelsn, iouv)                                         utils/metrics.py:224-265
// for every image of a batch in ONE launch, reading the padded NMS output rows where y5_nms_batched left them.
//
// process_batch, restated without the sort: the reference lists all (label, detection) pairs with the same class and
// IoU >= iouv[i], orders them by descending IoU, keeps the first row per detection, then (rows now ordered by detection
// index) the first row per label.  The first row per detection is that detection's best class-matched label l*(d), and
// that label does not depend on the threshold (if the best pair fails the threshold every other pair of d fails it too).
// So:  correct[d, i] = iou*(d) >= iouv[i]  and  no d' < d with l*(d') == l*(d) and iou*(d') >= iouv[i].
// Tie rule (equal IoU of one detection with two labels): the later label wins, which is what numpy's argsort()[::-1]
// gives for the short lists it sorts by insertion; longer lists are implementation-defined in the reference.
// One workgroup per image; labels are streamed through LDS in tiles, each lane owns detections d = lane, lane+256, ...
// IoU arithmetic follows box_iou (ultralytics.utils.metrics, call site utils/metrics.py:252) operation by operation in
// fp32; the file is built with -ffp-contract=off so the >= comparisons see the same bits as the reference's.
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "y5_common.h"
#include "y5_host.h"

namespace {
constexpr int MT = 256;  // threads per workgroup = label tile size

struct MatchParams {
  const float* det;       // (bs, max_det, ld_det) rows x1,y1,x2,y2,conf,cls,...
  const int* det_count;   // (bs) valid rows, or nullptr: every image has max_det rows
  const float* lab;       // (M, ld_lab) rows
  const float* scale;     // (bs, 5) gain, pad_x, pad_y, h0, w0 or nullptr
  const float* iouv;      // (niou)
  unsigned char* correct; // (bs, max_det, niou)
  float* predn;           // (bs, max_det, 4) or nullptr
  int bs, max_det, ld_det, M, ld_lab, img_col, cls_col, box_col, xywh, niou;
};

// scale_boxes with ratio_pad (utils/general.py:613-626 + clip_boxes :629-640)
__device__ inline void de_letterbox(float& x1, float& y1, float& x2, float& y2, const float* sc) {
  const float gain = sc[0], px = sc[1], py = sc[2], h0 = sc[3], w0 = sc[4];
  x1 = (x1 - px) / gain; x2 = (x2 - px) / gain;
  y1 = (y1 - py) / gain; y2 = (y2 - py) / gain;
  x1 = fminf(fmaxf(x1, 0.f), w0); x2 = fminf(fmaxf(x2, 0.f), w0);
  y1 = fminf(fmaxf(y1, 0.f), h0); y2 = fminf(fmaxf(y2, 0.f), h0);
}
}  // namespace

__global__ __launch_bounds__(256)
void y5_val_match_kernel(const MatchParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_lab = reinterpret_cast<float*>(smem);                    // [MT][5]: cls, x1, y1, x2, y2 (cls = NaN: not this image)
  int* s_best = reinterpret_cast<int*>(s_lab + MT * 5);             // [max_det] label row of the best pair, -1 none
  unsigned* s_mask = reinterpret_cast<unsigned*>(s_best + p.max_det);  // [max_det] bit i: iou* >= iouv[i]
  const int si = blockIdx.x, tid = threadIdx.x;
  int n = p.det_count ? p.det_count[si] : p.max_det;
  n = n < 0 ? 0 : (n > p.max_det ? p.max_det : n);
  const float* sc = p.scale ? p.scale + (size_t)si * 5 : nullptr;

  // this lane's detections: registers for up to 4 (max_det <= 1024)
  float bx1[4], by1[4], bx2[4], by2[4], bcls[4], biou[4];
  int bl[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + k * MT;
    bl[k] = -1; biou[k] = -1.f; bcls[k] = 0.f; bx1[k] = by1[k] = bx2[k] = by2[k] = 0.f;
    if (d < n) {
      const float* r = p.det + ((size_t)si * p.max_det + d) * p.ld_det;
      float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
      if (sc) de_letterbox(x1, y1, x2, y2, sc);
      bx1[k] = x1; by1[k] = y1; bx2[k] = x2; by2[k] = y2; bcls[k] = r[5];
      if (p.predn) {
        float* o = p.predn + ((size_t)si * p.max_det + d) * 4;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
      }
    }
  }

  for (int t0 = 0; t0 < p.M; t0 += MT) {
    __syncthreads();
    {
      const int t = t0 + tid;
      float c = __builtin_nanf("");
      float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
      if (t < p.M) {
        const float* r = p.lab + (size_t)t * p.ld_lab;
        if (p.img_col < 0 || r[p.img_col] == (float)si) {
          c = r[p.cls_col];
          const float a = r[p.box_col], b = r[p.box_col + 1], w = r[p.box_col + 2], h = r[p.box_col + 3];
          if (p.xywh) {  // xywh2xyxy (ultralytics.utils.ops; call site val.py:303)
            const float hw = w / 2, hh = h / 2;
            x1 = a - hw; y1 = b - hh; x2 = a + hw; y2 = b + hh;
          } else {
            x1 = a; y1 = b; x2 = w; y2 = h;
          }
          if (sc) de_letterbox(x1, y1, x2, y2, sc);
        }
      }
      float* o = s_lab + tid * 5;
      o[0] = c; o[1] = x1; o[2] = y1; o[3] = x2; o[4] = y2;
    }
    __syncthreads();
    const int tn = p.M - t0 < MT ? p.M - t0 : MT;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (tid + k * MT >= n) continue;
      const float area_d = (bx2[k] - bx1[k]) * (by2[k] - by1[k]);
      for (int j = 0; j < tn; ++j) {
        const float* l = s_lab + j * 5;
        if (!(l[0] == bcls[k])) continue;  // class mismatch or a label of another image (NaN)
        const float iw = fmaxf(fminf(l[3], bx2[k]) - fmaxf(l[1], bx1[k]), 0.f);
        const float ih = fmaxf(fminf(l[4], by2[k]) - fmaxf(l[2], by1[k]), 0.f);
        const float inter = iw * ih;
        const float area_l = (l[3] - l[1]) * (l[4] - l[2]);
        const float iou = inter / (area_l + area_d - inter + 1e-7f);
        if (iou >= biou[k]) { biou[k] = iou; bl[k] = t0 + j; }
      }
    }
  }

#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + k * MT;
    if (d < n) {
      unsigned m = 0;
      if (bl[k] >= 0)
        for (int i = 0; i < p.niou; ++i) m |= (biou[k] >= p.iouv[i] ? 1u : 0u) << i;
      s_best[d] = bl[k];
      s_mask[d] = m;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + k * MT;
    if (d >= p.max_det) continue;
    unsigned ok = 0;
    if (d < n) {
      ok = s_mask[d];
      const int l = s_best[d];
      for (int e = 0; e < d && ok; ++e)
        if (s_best[e] == l) ok &= ~s_mask[e];
    }
    unsigned char* o = p.correct + ((size_t)si * p.max_det + d) * p.niou;
    for (int i = 0; i < p.niou; ++i) o[i] = (ok >> i) & 1u;  // rows past the count are written as 0
  }
}

// scale_boxes (utils/general.py:613-626) in place on the padded NMS rows of every image: detect.py:248, models/common.py:941
__global__ __launch_bounds__(256)
void y5_scale_boxes_kernel(float* __restrict__ det, int ld, int max_det, const int* __restrict__ count, const float* __restrict__ scale, int do_round) {
  const int si = blockIdx.y;
  const int d = blockIdx.x * 256 + threadIdx.x;
  int n = count ? count[si] : max_det;
  n = n > max_det ? max_det : n;
  if (d >= n) return;
  float* r = det + ((size_t)si * max_det + d) * ld;
  float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
  de_letterbox(x1, y1, x2, y2, scale + (size_t)si * 5);
  if (do_round) { x1 = rintf(x1); y1 = rintf(y1); x2 = rintf(x2); y2 = rintf(y2); }  // torch.round: half to even (detect.py:248)
  r[0] = x1; r[1] = y1; r[2] = x2; r[3] = y2;
}

extern "C" int y5_scale_boxes_batch(float* det, int ld_det, int max_det, const int* det_count, int bs, const float* scale, int do_round,
                                    void* stream_) {
  if (!det || !scale) return y5_fail(Y5_ERR_BAD_ARG, "scale_boxes_batch: null pointer");
  if (bs < 1 || bs > 65535 || max_det < 1 || ld_det < 4) return y5_fail(Y5_ERR_BAD_ARG, "scale_boxes_batch: need 1 <= bs <= 65535, max_det >= 1, ld_det >= 4");
  hipLaunchKernelGGL(y5_scale_boxes_kernel, dim3((max_det + 255) / 256, bs), dim3(256), 0, static_cast<hipStream_t>(stream_), det, ld_det, max_det,
                     det_count, scale, do_round);
  return y5_check_launch("y5_scale_boxes_batch");
}

extern "C" int y5_val_match(const float* det, int ld_det, int max_det, const int* det_count, int bs, const float* labels, int ld_lab,
                            int nlabels, int img_col, int cls_col, int box_col, int xywh, const float* scale, const float* iouv, int niou,
                            unsigned char* correct, float* predn, void* stream_) {
  if (!det || !iouv || !correct || (nlabels > 0 && !labels)) return y5_fail(Y5_ERR_BAD_ARG, "val_match: null pointer");
  if (bs < 1 || max_det < 1 || max_det > 4 * MT || ld_det < 6 || niou < 1 || niou > 32 || nlabels < 0)
    return y5_fail(Y5_ERR_BAD_ARG, "val_match: need bs >= 1, 1 <= max_det <= 1024, ld_det >= 6, 1 <= niou <= 32");
  if (nlabels > 0 && (cls_col < 0 || box_col < 0 || cls_col >= ld_lab || box_col + 4 > ld_lab || img_col >= ld_lab))
    return y5_fail(Y5_ERR_BAD_ARG, "val_match: label columns outside the row");
  MatchParams p{};
  p.det = det; p.det_count = det_count; p.lab = labels; p.scale = scale; p.iouv = iouv; p.correct = correct; p.predn = predn;
  p.bs = bs; p.max_det = max_det; p.ld_det = ld_det; p.M = nlabels; p.ld_lab = ld_lab; p.img_col = img_col; p.cls_col = cls_col;
  p.box_col = box_col; p.xywh = xywh; p.niou = niou;
  const size_t lds = (size_t)MT * 5 * sizeof(float) + (size_t)max_det * 8;
  hipLaunchKernelGGL(y5_val_match_kernel, dim3(bs), dim3(MT), lds, static_cast<hipStream_t>(stream_), p);
  return y5_check_launch("y5_val_match");
}
