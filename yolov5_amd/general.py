"""Post-processing entry points with the reference's names and signatures (utils/general.py):
`non_max_suppression` :658-767 (HIP: y5_nms_batched), `scale_boxes` :613-626, `xyxy2xywh` :574-581, plus the
un-vendored helpers the reference imports from `ultralytics.utils.ops` (xywh2xyxy, clip_boxes, make_divisible).
"""
from __future__ import annotations

import os

import ctypes as C
import logging
import math

import torch

from . import _lib

LOGGER = logging.getLogger("yolov5_amd")


def make_divisible(x, divisor):
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def xyxy2xywh(x):
    """utils/general.py:574-581."""
    y = x.clone()
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def xywh2xyxy(x):
    y = x.clone()
    y[..., :2] = x[..., :2] - x[..., 2:4] / 2
    y[..., 2:4] = x[..., :2] + x[..., 2:4] / 2
    return y


def clip_boxes(boxes, shape):
    boxes[..., 0].clamp_(0, shape[1])
    boxes[..., 1].clamp_(0, shape[0])
    boxes[..., 2].clamp_(0, shape[1])
    boxes[..., 3].clamp_(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """utils/general.py:613-626: de-letterbox xyxy boxes in place (n is at most max_det: host-side torch ops)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    boxes[..., [0, 2]] -= pad[0]
    boxes[..., [1, 3]] -= pad[1]
    boxes[..., :4] /= gain
    clip_boxes(boxes, img0_shape)
    return boxes


def scale_boxes_batch(img1_shape, out, counts, img0_shapes, ratio_pads=None, round_=False):
    """`scale_boxes` (utils/general.py:613-626) for every image of a padded NMS result in one launch, in place: out (bs, max_det,
    6+nm) fp32 / counts (bs) int32 as returned by non_max_suppression(..., padded=True); img0_shapes[i] = (h0, w0);
    ratio_pads[i] = ((gain, gain), (pad_x, pad_y)) or None (computed from the shapes, :615-617).  round_: detect.py:248."""
    if not _lib.accepts(out) or out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("scale_boxes_batch needs the contiguous fp32 GPU buffer of non_max_suppression(..., padded=True)")
    bs, max_det, ld = out.shape
    rows = []
    for i, s0 in enumerate(img0_shapes):
        rp = ratio_pads[i] if ratio_pads is not None else None
        if rp is None:
            gain = min(img1_shape[0] / s0[0], img1_shape[1] / s0[1])
            pad = (img1_shape[1] - s0[1] * gain) / 2, (img1_shape[0] - s0[0] * gain) / 2
        else:
            gain, pad = rp[0][0], rp[1]
        rows.append([gain, pad[0], pad[1], s0[0], s0[1]])
    sc = torch.tensor(rows, dtype=torch.float32).to(out.device)
    lib = _lib.lib()
    rc = lib.y5_scale_boxes_batch(C.c_void_p(out.data_ptr()), ld, max_det, C.c_void_p(counts.data_ptr()) if counts is not None else None, bs,
                                  C.c_void_p(sc.data_ptr()), int(round_), _lib.stream(out.device))
    _lib.check(rc, lib)
    return out


_nms_ws = {}


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, nm=0, padded=False):
    """utils/general.py:658-767 on the GPU: one batched HIP kernel chain for all images (filter -> LDS bitonic sort
    -> greedy IoU with kept boxes in LDS), ONE device->host sync (the per-image counts) instead of >= 3 per image.

    Returns list of (k, 6+nm) fp32 tensors [x1, y1, x2, y2, conf, cls, (mask coefficients)] per image, rows in
    descending confidence; with padded=True the device buffers themselves, (out (bs, max_det, 6+nm), counts (bs) int32),
    without any host sync (input of metrics.match_batch / ValStats).  Deliberate differences from the reference, both documented in DESIGN.md:
      * arithmetic is fp32 on the fp32 value of every element also for half inputs (the reference's half path
        overflows the class offset cls*7680 for cls >= 9); output rows are always fp32;
      * no wall-clock `time_limit` (general.py:692,763-765) -- results never depend on timing;
      * ties in confidence are ordered by the lower candidate index (the reference's argsort is unstable).
    """
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):  # model in validation mode: (inference_out, loss_out)
        prediction = prediction[0]
    if not _lib.accepts(prediction):
        raise RuntimeError("yolov5_amd.non_max_suppression needs a GPU tensor (no CPU path)")
    if prediction.dtype not in (torch.float16, torch.float32):
        prediction = prediction.float()
    if labels and any(len(lb) for lb in labels):
        # general.py:706-712 (val.py --save-hybrid autolabelling): every image's a-priori labels [cls, x, y, w, h] (pixels) join ITS candidates as
        # rows with objectness 1 and a one-hot class.  Here they are appended as extra prediction rows behind the model's (image slots without a
        # label keep objectness 0 and never become candidates): same candidate order as the reference's torch.cat((x, v), 0), so the same tie
        # rule.  fp32 like the reference's concatenation of a half tensor with the float label rows.
        bs0, _, no0 = prediction.shape
        lmax = max(len(lb) for lb in labels)
        extra = torch.zeros((bs0, lmax, no0), dtype=torch.float32, device=prediction.device)
        for xi, lb in enumerate(labels):
            if len(lb):
                lb = torch.as_tensor(lb, dtype=torch.float32, device=prediction.device)
                extra[xi, :len(lb), :4] = lb[:, 1:5]
                extra[xi, :len(lb), 4] = 1.0
                extra[xi, torch.arange(len(lb), device=prediction.device), lb[:, 0].long() + 5] = 1.0
        prediction = torch.cat((prediction.float(), extra), 1)
    prediction = prediction.contiguous()
    # objectness plane the engine wrote beside this very tensor (engine.Engine._tag_hint): used only while `prediction` is still the object the
    # engine's LATEST forward returned, unmodified -- any copy, cast, slice, in-place edit or later forward drops it and the filter reads the rows themselves
    hint = None
    tag = getattr(prediction, "_y5_obj_hint", None)
    if tag is not None and not prediction.is_inference():
        h, ver, ptr0, state, k, seq = tag
        # (state[k] == seq: no later forward of that engine has written into this plane -- the engine owns two and alternates)
        if (state[k] == seq and ver == prediction._version and ptr0 == prediction.data_ptr() and h.dtype == prediction.dtype
                and h.device == prediction.device and tuple(h.shape) == tuple(prediction.shape[:2]) and h.is_contiguous()):
            hint = h
    lib = _lib.lib()
    bs, n, no = prediction.shape
    nc = no - nm - 5
    flags = (_lib.NMS_MULTI_LABEL if (multi_label and nc > 1) else 0) | (_lib.NMS_AGNOSTIC if agnostic else 0)
    max_nms = 30000
    dev = prediction.device
    # the scratch workspace is private to (shape, stream): a DetectPipeline's side-stream NMS and a main-stream call of the same shape
    # may be in flight together and must not share candidate lists
    sid = torch.cuda.current_stream(dev).cuda_stream if prediction.is_cuda else 0
    key = (bs, n, no, nm, flags, max_det, str(dev), sid)
    ws = _nms_ws.get(key)
    if ws is None:
        nbytes = lib.y5_nms_workspace_bytes(bs, n, no, nm, flags, max_nms)
        # bounded by BYTES as well as by count: a val.py-shaped call (bs 64, n * nc ~ 2 M keys) needs > 1 GiB of scratch, a few of
        # those beside detect-shaped ones must not pin several GiB -- oldest entries go until the cache fits the budget (the newest always stays)
        budget = int(os.environ.get("Y5_NMS_WS_CACHE_MB", "1536")) << 20
        while _nms_ws and (len(_nms_ws) >= 4 or sum(v[1] for v in _nms_ws.values()) + nbytes > budget):
            _nms_ws.pop(next(iter(_nms_ws)))
        ws = _nms_ws[key] = (_lib.workspace(nbytes, dev), nbytes)
    out = torch.empty((bs, max_det, 6 + nm), dtype=torch.float32, device=dev)
    cnt = torch.empty((bs,), dtype=torch.int32, device=dev)
    cls_t = None
    if classes is not None:
        cls_t = torch.tensor(list(classes), dtype=torch.int32, device=dev)
    dt = _lib.Y5_F16 if prediction.dtype == torch.float16 else _lib.Y5_F32
    rc = lib.y5_nms_batched_hint(C.c_void_p(prediction.data_ptr()), dt, bs, n, no, nm, float(conf_thres), float(iou_thres), int(max_det),
                                 max_nms, 7680.0, flags, C.c_void_p(cls_t.data_ptr()) if cls_t is not None else None,
                                 0 if cls_t is None else cls_t.numel(), C.c_void_p(out.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                 C.c_void_p(ws[0].data_ptr()), ws[1], C.c_void_p(hint.data_ptr()) if hint is not None else None, _lib.stream(dev))
    _lib.check(rc, lib)
    if padded:
        return out, cnt
    counts = cnt.tolist()  # the single D2H sync
    # per-image views of the padded buffer through ONE split call (sizes c0, max_det - c0, c1, ...: the even parts are the results) --
    # 64 Python-level slicings cost ~100 us per batch, a third of the kernels' time
    sizes = [v for c in counts for v in (c, max_det - c)]
    return list(out.view(bs * max_det, 6 + nm).split(sizes)[::2])
