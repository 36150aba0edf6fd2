// C-ABI launcher for batched NMS (nms_kernels.h).  Built with -ffp-contract=off (bit-exact selection).
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "nms_kernels.h"
#include "y5_host.h"

namespace {
struct Layout { size_t off_count, off_keys, off_cls, off_gbox, off_rows, off_hist, off_thr, off_keys2, total; long long cap, cap_pad, gcap, cap2; };
Layout layout(int bs, int n, int no, int nm, int flags, int max_nms) {
  Layout L{};
  const int nc = no - 5 - nm;
  L.cap = (flags & Y5_NMS_MULTI_LABEL) && nc > 1 ? (long long)n * nc : n;
  long long p2 = 64;
  while (p2 < L.cap) p2 <<= 1;
  L.cap_pad = p2;
  size_t o = 0;
  L.off_count = o; o += ((size_t)bs * 8 + 255) & ~(size_t)255;  // count[bs] + rcount[bs] (hint path)
  L.off_keys = o; o += ((size_t)bs * L.cap_pad * 8 + 255) & ~(size_t)255;
  L.off_cls = o; o += ((size_t)bs * n + 255) & ~(size_t)255;
  L.gcap = L.cap < max_nms ? L.cap : max_nms;
  L.gcap = (L.gcap + 63) / 64 * 64;  // whole chunks of 64 candidates (field planes, nms_kernels.h)
  if (L.gcap < 64) L.gcap = 64;
  L.off_gbox = o; o += ((size_t)bs * L.gcap * Y5_NMS_REC * 4 + 255) & ~(size_t)255;
  L.off_rows = o; o += ((size_t)bs * n * 4 + 255) & ~(size_t)255;  // hint path: rows the objectness plane could not exclude
  // pruning stage (nms_kernels.h K1c), only where a candidate list can exceed max_nms: histogram, thresholds + counts, compaction buffer of 4 x max_nms keys
  L.cap2 = L.cap > max_nms ? (4LL * max_nms + 63) / 64 * 64 : 0;
  if (L.cap2 > L.cap) L.cap2 = L.cap;
  L.off_hist = o; o += L.cap2 ? ((size_t)bs * Y5_NMS_BINS * 4 + 255) & ~(size_t)255 : 0;
  L.off_thr = o; o += L.cap2 ? ((size_t)bs * 8 + 255) & ~(size_t)255 : 0;
  L.off_keys2 = o; o += ((size_t)bs * L.cap2 * 8 + 255) & ~(size_t)255;
  L.total = o;
  return L;
}
}  // namespace

extern "C" size_t y5_nms_workspace_bytes(int bs, int n, int no, int nm, int flags, int max_nms) {
  return layout(bs, n, no, nm, flags, max_nms > 0 ? max_nms : 30000).total;
}

extern "C" int y5_nms_batched_hint(const void* pred, int dt, int bs, int n, int no, int nm, float conf_thres, float iou_thres, int max_det,
                                   int max_nms, float max_wh, int flags, const int* classes, int nclasses, float* out, int* out_count,
                                   void* ws, size_t ws_bytes, const void* obj_hint, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int nc = no - 5 - nm;
  if (!pred || !out || !out_count || !ws || bs <= 0 || n <= 0 || nc < 1 || nc > 256 || nm < 0)
    return y5_fail(Y5_ERR_BAD_ARG, "nms: bad args");
  if (!(conf_thres >= 0.f && conf_thres <= 1.f)) return y5_fail(Y5_ERR_BAD_ARG, "nms: Invalid Confidence threshold, valid values are between 0.0 and 1.0");
  if (!(iou_thres >= 0.f && iou_thres <= 1.f)) return y5_fail(Y5_ERR_BAD_ARG, "nms: Invalid IoU, valid values are between 0.0 and 1.0");
  if (max_det < 1 || max_det > Y5_NMS_MAX_DET_CAP) return y5_fail(Y5_ERR_UNSUPPORTED, "nms: max_det out of range [1,4096]");
  if (dt != Y5_F16 && dt != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "nms: dtype");
  if ((flags & Y5_NMS_MULTI_LABEL) && nc <= 1) flags &= ~Y5_NMS_MULTI_LABEL;  // general.py:693
  if (max_nms < 1) return y5_fail(Y5_ERR_BAD_ARG, "nms: max_nms must be positive");
  const Layout L = layout(bs, n, no, nm, flags, max_nms);
  if (ws_bytes < L.total || ((uintptr_t)ws & 255)) return y5_fail(Y5_ERR_WORKSPACE, "nms: workspace too small or misaligned");

  Y5NmsParams p{};
  p.pred = pred; p.obj_hint = obj_hint; p.bs = bs; p.n = n; p.no = no; p.nc = nc; p.nm = nm;
  p.conf_thres = conf_thres; p.iou_thres = iou_thres; p.max_wh = max_wh;
  p.max_det = max_det; p.max_nms = max_nms; p.flags = flags;
  p.classes = nclasses > 0 ? classes : nullptr; p.nclasses = nclasses;
  p.out = out; p.out_count = out_count;
  char* w = static_cast<char*>(ws);
  p.count = reinterpret_cast<int*>(w + L.off_count);
  p.rcount = p.count + bs;
  p.rows = reinterpret_cast<int*>(w + L.off_rows);
  p.keys = reinterpret_cast<unsigned long long*>(w + L.off_keys);
  p.best_cls = reinterpret_cast<unsigned char*>(w + L.off_cls);
  p.gbox = reinterpret_cast<float*>(w + L.off_gbox);
  p.cap = L.cap; p.cap_pad = L.cap_pad; p.gcap = L.gcap;
  p.hist = reinterpret_cast<int*>(w + L.off_hist); p.thr = reinterpret_cast<int*>(w + L.off_thr);
  p.keys2 = reinterpret_cast<unsigned long long*>(w + L.off_keys2); p.cap2 = L.cap2;

  if (hipMemsetAsync(p.count, 0, (size_t)bs * 8, st) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "nms: memset failed");
  const dim3 fg((unsigned)((n + 255) / 256), (unsigned)bs), fb(256);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)y5_nms_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Y5_NMS_SORT_LDS_KEYS * 8);
    hipFuncSetAttribute((const void*)y5_nms_sort_windows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Y5_NMS_SORT_LDS_KEYS * 8);
    hipFuncSetAttribute((const void*)y5_nms_greedy_kernel<half_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void*)y5_nms_greedy_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr = true;
  }
  size_t greedy_lds = (((size_t)max_det * 6 + 3) & ~(size_t)3) * 4 + (size_t)Y5_NMS_RING * Y5_NMS_REC * 64 * 4 + 64 * Y5_NMS_GREEDY_WAVES + Y5_NMS_GREEDY_WAVES * 8 + 16;
  {
    // per-(class, wave) kept lists of the greedy kernel: when NMS is per class and heads + links fit beside the kept boxes
    const size_t lists = ((size_t)nc * Y5_NMS_GREEDY_WAVES + (size_t)max_det) * 4;
    p.cls_lists = !(flags & Y5_NMS_AGNOSTIC) && greedy_lds + lists <= 128 * 1024;
    if (p.cls_lists) greedy_lds += lists;
  }
  {
    // rows per workgroup of the LDS-staged filter: as many as fit 64 KiB (multiple of 64); unstaged fallback for very wide rows
    const int es = dt == Y5_F16 ? 2 : 4;
    int rows = (int)(65536 / ((long long)no * es)) / 64 * 64;
    if (rows > 256) rows = 256;
    if (p.obj_hint) {
      // objectness plane beside z: a scan of the plane lists the rows it cannot exclude (wave-aggregated compaction), then one thread per
      // listed row decides it on `pred` itself (grid-stride over the list: 8 workgroups of 256 threads per image)
      const dim3 rg(8, (unsigned)bs);
      const size_t rl = (size_t)256 * no * es;  // the workgroup's rows
      if (rl > 64 * 1024) {  // very wide rows: the plain filters below (obj_hint unused)
        p.obj_hint = nullptr;
      } else if (dt == Y5_F16) {
        hipLaunchKernelGGL((y5_nms_hint_scan_kernel<half_t>), fg, fb, 0, st, p);
        hipLaunchKernelGGL((y5_nms_hint_rows_kernel<half_t>), rg, dim3(256), rl, st, p);
      } else {
        hipLaunchKernelGGL((y5_nms_hint_scan_kernel<float>), fg, fb, 0, st, p);
        hipLaunchKernelGGL((y5_nms_hint_rows_kernel<float>), rg, dim3(256), rl, st, p);
      }
    }
    if (p.obj_hint) {
      // (filtered above through the objectness plane)
    } else if (rows >= 64) {
      const dim3 sg((unsigned)((n + rows - 1) / rows), (unsigned)bs);
      const size_t lds = (size_t)rows * no * es;
      if (dt == Y5_F16) hipLaunchKernelGGL((y5_nms_filter_kernel<half_t, true>), sg, dim3(rows), lds, st, p);
      else hipLaunchKernelGGL((y5_nms_filter_kernel<float, true>), sg, dim3(rows), lds, st, p);
    } else if (dt == Y5_F16) {
      hipLaunchKernelGGL((y5_nms_filter_kernel<half_t, false>), fg, fb, 0, st, p);
    } else {
      hipLaunchKernelGGL((y5_nms_filter_kernel<float, false>), fg, fb, 0, st, p);
    }
  }
  if (L.cap2 > 0) {  // a list can be longer than max_nms: keep only the keys that can be among the max_nms best before sorting
    if (hipMemsetAsync(p.hist, 0, (size_t)bs * Y5_NMS_BINS * 4, st) != hipSuccess) return y5_fail(Y5_ERR_RUNTIME, "nms: memset failed");
    const dim3 hg(64, (unsigned)bs);
    hipLaunchKernelGGL(y5_nms_hist_kernel, hg, dim3(256), Y5_NMS_BINS * 4, st, p);
    hipLaunchKernelGGL(y5_nms_select_kernel, dim3((unsigned)bs), dim3(256), 256 * 4, st, p);
    hipLaunchKernelGGL(y5_nms_compact_kernel, hg, dim3(256), 16, st, p);
    hipLaunchKernelGGL(y5_nms_adopt_kernel, dim3(16, (unsigned)bs), dim3(256), 0, st, p);
    hipLaunchKernelGGL(y5_nms_adopt_count_kernel, dim3((unsigned)((bs + 255) / 256)), dim3(256), 0, st, p);
  }
  if (L.cap_pad > Y5_NMS_SORT_LDS_KEYS) {  // lists that can exceed one LDS window: the window-local merge sizes in parallel (early exit per image otherwise)
    const long long most = L.cap2 > 0 ? (L.cap2 > max_nms ? L.cap2 : max_nms) : L.cap;   // longest list the sort can see (pruned, or the overflow fallback)
    long long wp2 = 64;
    while (wp2 < (L.cap2 > 0 ? L.cap : most)) wp2 <<= 1;
    hipLaunchKernelGGL(y5_nms_sort_windows_kernel, dim3((unsigned)(wp2 / Y5_NMS_SORT_LDS_KEYS), (unsigned)bs), dim3(1024), Y5_NMS_SORT_LDS_KEYS * 8, st, p);
  }
  hipLaunchKernelGGL(y5_nms_sort_kernel, dim3((unsigned)bs), dim3(1024), Y5_NMS_SORT_LDS_KEYS * 8, st, p);
  {
    const dim3 gg((unsigned)((L.gcap + 255) / 256), (unsigned)bs);
    if (dt == Y5_F16) hipLaunchKernelGGL((y5_nms_gather_kernel<half_t>), gg, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((y5_nms_gather_kernel<float>), gg, dim3(256), 0, st, p);
  }
  if (dt == Y5_F16) hipLaunchKernelGGL((y5_nms_greedy_kernel<half_t>), dim3((unsigned)bs), dim3(64 * Y5_NMS_GREEDY_WAVES), greedy_lds, st, p);
  else hipLaunchKernelGGL((y5_nms_greedy_kernel<float>), dim3((unsigned)bs), dim3(64 * Y5_NMS_GREEDY_WAVES), greedy_lds, st, p);
  return y5_check_launch("y5_nms_batched");
}

extern "C" int y5_nms_batched(const void* pred, int dt, int bs, int n, int no, int nm, float conf_thres, float iou_thres, int max_det,
                              int max_nms, float max_wh, int flags, const int* classes, int nclasses, float* out, int* out_count,
                              void* ws, size_t ws_bytes, void* stream_) {
  return y5_nms_batched_hint(pred, dt, bs, n, no, nm, conf_thres, iou_thres, max_det, max_nms, max_wh, flags, classes, nclasses, out, out_count, ws, ws_bytes,
                             nullptr, stream_);
}
