"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (yolov5_amd/).

Restatements of the THIRD-PARTY arithmetic the reference hot path calls but does not vendor
(SURVEY.md section 8c).  Dependencies and pins, from /root/reference:

  * `ultralytics>=8.4.118`  (requirements.txt:16, pyproject.toml:78)  -- NOT installed here
  * `torchvision>=0.9.0`    (requirements.txt:15)                     -- NOT installed here

PARITY UNPINNED: the reference's own tests hold no golden vectors for any of these functions
(tests/ covers SSRF / shell-injection / Flask only), and neither wheel can be installed in this
container (no network).  Each function below restates the published upstream algorithm; parity is
anchored on the reference's call sites, cited per function.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def xywh2xyxy(x):
    """ultralytics.utils.ops.xywh2xyxy; call site utils/general.py:722 (NMS box conversion).

    (cx, cy, w, h) -> (cx - w/2, cy - h/2, cx + w/2, cy + h/2), computed in the input dtype.
    """
    y = x.clone() if isinstance(x, torch.Tensor) else x.copy()
    xy = x[..., :2]
    wh = x[..., 2:4] / 2
    y[..., :2] = xy - wh
    y[..., 2:4] = xy + wh
    return y


def clip_boxes(boxes, shape):
    """ultralytics.utils.ops.clip_boxes; call site utils/general.py:625 (scale_boxes).

    In-place clamp of xyxy boxes to x in [0, w], y in [0, h]; `shape` is (h, w).
    """
    if isinstance(boxes, torch.Tensor):
        boxes[..., 0].clamp_(0, shape[1])
        boxes[..., 1].clamp_(0, shape[0])
        boxes[..., 2].clamp_(0, shape[1])
        boxes[..., 3].clamp_(0, shape[0])
    else:
        boxes[..., [0, 2]] = boxes[..., [0, 2]].clip(0, shape[1])
        boxes[..., [1, 3]] = boxes[..., [1, 3]].clip(0, shape[0])
    return boxes


def make_divisible(x, divisor):
    """ultralytics.utils.ops.make_divisible; call sites models/yolo.py:423,441."""
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def smooth_bce(eps=0.1):
    """ultralytics.utils.metrics.smooth_bce; call site utils/loss.py:117.  -> (positive, negative) targets."""
    return 1.0 - 0.5 * eps, 0.5 * eps


def bbox_iou(box1, box2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """ultralytics.utils.metrics.bbox_iou; call site utils/loss.py:153 (`CIoU=True`, xywh boxes).

    box1 (n,4), box2 (n,4) -> (n,1).  alpha is computed under no_grad (detached), as upstream.
    """
    if xywh:
        (x1, y1, w1, h1), (x2, y2, w2, h2) = box1.chunk(4, -1), box2.chunk(4, -1)
        w1_, h1_, w2_, h2_ = w1 / 2, h1 / 2, w2 / 2, h2 / 2
        b1_x1, b1_x2, b1_y1, b1_y2 = x1 - w1_, x1 + w1_, y1 - h1_, y1 + h1_
        b2_x1, b2_x2, b2_y1, b2_y2 = x2 - w2_, x2 + w2_, y2 - h2_, y2 + h2_
    else:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1.chunk(4, -1)
        b2_x1, b2_y1, b2_x2, b2_y2 = box2.chunk(4, -1)
        w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
        w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps

    inter = (b1_x2.minimum(b2_x2) - b1_x1.maximum(b2_x1)).clamp_(0) * (
        b1_y2.minimum(b2_y2) - b1_y1.maximum(b2_y1)
    ).clamp_(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if CIoU or DIoU or GIoU:
        cw = b1_x2.maximum(b2_x2) - b1_x1.minimum(b2_x1)
        ch = b1_y2.maximum(b2_y2) - b1_y1.minimum(b2_y1)
        if CIoU or DIoU:
            c2 = cw.pow(2) + ch.pow(2) + eps
            rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2).pow(2) + (b2_y1 + b2_y2 - b1_y1 - b1_y2).pow(2)) / 4
            if CIoU:
                v = (4 / math.pi**2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
                with torch.no_grad():
                    alpha = v / (v - iou + (1 + eps))
                return iou - (rho2 / c2 + v * alpha)
            return iou - rho2 / c2
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area
    return iou


def box_iou(box1, box2, eps=1e-7):
    """ultralytics.utils.metrics.box_iou; pairwise IoU (n,4)x(m,4)->(n,m); call site utils/metrics.py:252."""
    (a1, a2), (b1, b2) = box1.float().unsqueeze(1).chunk(2, 2), box2.float().unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp_(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + eps)


def smooth(y, f=0.05):
    """ultralytics.utils.metrics.smooth; call site utils/metrics.py:91.  Box filter of fraction f over y with edge padding:
    nf = odd number of taps closest to 2*f*len(y); y is extended by nf//2 copies of its first / last value, then averaged."""
    import numpy as np

    nf = round(len(y) * f * 2) // 2 + 1
    pad = np.ones(nf // 2)
    yp = np.concatenate((pad * y[0], y, pad * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def cv2_resize(src, dsize, dst=None, fx=0, fy=0, interpolation=1):
    """cv2.resize(im, (w, h), interpolation=cv2.INTER_LINEAR) for 8-bit images; call site utils/augmentations.py:110.
    PARITY UNPINNED (opencv-python is absent): restates OpenCV's published algorithm, modules/imgproc/src/resize.cpp --
      scale = 1 / (dsize / ssize) in double;  an exact 2x down-scale in x and y is re-routed to INTER_AREA whose fast path
      is the rounded mean of the 2x2 block;  otherwise for every destination index d: f = float((d + 0.5) * scale - 0.5),
      s = floor(f), f -= s; along x  s < 0 -> (0, 0)  and  s >= w - 1 -> (w - 1, 0); along y the two rows are clamped to the
      image and f is kept; weights = short(rint(w * 2048)) (half to even); horizontal pass h = S[s] * a0 + S[s + 1] * a1 in
      int32, vertical pass out = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2."""
    import numpy as np

    assert interpolation == 1 and src.dtype == np.uint8, "only INTER_LINEAR on uint8 is restated"
    squeeze = src.ndim == 2
    s3 = src[:, :, None] if squeeze else src
    sh, sw = s3.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    scale_x, scale_y = 1.0 / (dw / sw), 1.0 / (dh / sh)
    eps = np.finfo(np.float64).eps
    if abs(scale_x - 2.0) < eps and abs(scale_y - 2.0) < eps:
        a = s3.astype(np.int32)
        out = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
        out = out.astype(np.uint8)
        return out[:, :, 0] if squeeze else out

    def taps(n_dst, scale, n_src, zero_outside):
        d = np.arange(n_dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        if zero_outside:
            lo, hi = s < 0, s >= n_src - 1
            f[lo] = 0.0; s[lo] = 0
            f[hi] = 0.0; s[hi] = n_src - 1
        w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int32)
        w1 = np.rint(f * np.float32(2048.0)).astype(np.int32)
        return np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1), w0, w1

    x0, x1, a0, a1 = taps(dw, scale_x, sw, True)
    y0, y1, b0, b1 = taps(dh, scale_y, sh, False)
    a = s3.astype(np.int32)
    h = a[:, x0] * a0[None, :, None] + a[:, x1] * a1[None, :, None]  # (sh, dw, c)
    out = (((b0[:, None, None] * (h[y0] >> 4)) >> 16) + ((b1[:, None, None] * (h[y1] >> 4)) >> 16) + 2) >> 2
    out = out.astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def cv2_copy_make_border(src, top, bottom, left, right, borderType=0, value=0):
    """cv2.copyMakeBorder(..., cv2.BORDER_CONSTANT, value=color); call site utils/augmentations.py:113."""
    import numpy as np

    assert borderType == 0
    h, w = src.shape[:2]
    val = np.asarray(value, dtype=src.dtype)
    val = val[: src.shape[2]] if (src.ndim == 3 and val.ndim) else val
    out = np.empty((h + top + bottom, w + left + right) + src.shape[2:], dtype=src.dtype)
    out[...] = val
    out[top:top + h, left:left + w] = src
    return out


def initialize_weights(model):
    """ultralytics.utils.torch_utils.initialize_weights; call site models/yolo.py:259.

    BatchNorm2d: eps=1e-3, momentum=0.03; activations inplace; conv weights keep torch default init.
    """
    for m in model.modules():
        t = type(m)
        if t is nn.Conv2d:
            pass
        elif t is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif t in {nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU}:
            m.inplace = True


def scale_img(img, ratio=1.0, same_shape=False, gs=32):
    """ultralytics.utils.torch_utils.scale_img; call site models/yolo.py:276 (TTA only)."""
    if ratio == 1.0:
        return img
    h, w = img.shape[2:]
    s = (int(h * ratio), int(w * ratio))
    img = F.interpolate(img, size=s, mode="bilinear", align_corners=False)
    if not same_shape:
        h, w = (math.ceil(x * ratio / gs) * gs for x in (h, w))
    return F.pad(img, [0, w - s[1], 0, h - s[0]], value=0.447)


def copy_attr(a, b, include=(), exclude=()):
    """ultralytics.utils.torch_utils.copy_attr; call site models/common.py:859 (AutoShape)."""
    for k, v in b.__dict__.items():
        if (len(include) and k not in include) or k.startswith("_") or k in exclude:
            continue
        setattr(a, k, v)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms; call site utils/general.py:750.

    Greedy NMS as in torchvision's CPU kernel (torchvision/csrc/ops/cpu/nms_kernel.cpp): candidates in
    descending-score order (stable: ties keep the lower input index first); walking i in that order, a
    non-suppressed i is kept and suppresses every later j whose
        inter / (area_i + area_j - inter)  >  iou_threshold      (strict; no +1, no eps)
    with inter = max(0, xx2-xx1) * max(0, yy2-yy1) and area = (x2-x1)*(y2-y1), all in the box dtype.
    Returns int64 indices into the input, in descending-score order.
    """
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().cpu().numpy()
    s = scores.detach().cpu().numpy()
    import numpy as np

    order = np.argsort(-s.astype(np.float64), kind="stable")
    bs = b[order]
    x1, y1, x2, y2 = bs[:, 0], bs[:, 1], bs[:, 2], bs[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    n = bs.shape[0]
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    zero = b.dtype.type(0)
    thr = b.dtype.type(iou_threshold)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(order[i])
        if i + 1 < n:
            xx1 = np.maximum(x1[i], x1[i + 1:])
            yy1 = np.maximum(y1[i], y1[i + 1:])
            xx2 = np.minimum(x2[i], x2[i + 1:])
            yy2 = np.minimum(y2[i], y2[i + 1:])
            w = np.maximum(zero, xx2 - xx1)
            h = np.maximum(zero, yy2 - yy1)
            inter = w * h
            with np.errstate(divide="ignore", invalid="ignore"):
                ovr = inter / (areas[i] + areas[i + 1:] - inter)
            suppressed[i + 1:] |= ovr > thr
    return torch.as_tensor(np.asarray(keep, dtype=np.int64), device=boxes.device)
