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


# --------------------------------------------------------------------------------------------------------------------
# OpenCV primitives of the training augmentations (utils/augmentations.py:69-83 augment_hsv, :118-190 random_perspective).
# PARITY UNPINNED: opencv-python (requirements.txt:9) is neither in the tree nor installed; these restate the published 8-bit
# algorithms of modules/imgproc (imgwarp.cpp: warpAffine -> remap with INTER_BITS = 5 fixed-point coordinates and 15-bit bilinear
# weights; color_hsv: the integer RGB->HSV of RGB2HSV_b and the float HSV->RGB of HSV2RGB_b).
# --------------------------------------------------------------------------------------------------------------------
COLOR_BGR2HSV, COLOR_HSV2BGR = 40, 54


def cv2_get_rotation_matrix_2d(center, angle, scale):
    """cv2.getRotationMatrix2D; call site utils/augmentations.py:143.  angle in degrees, positive = counter-clockwise (image y down)."""
    import numpy as np

    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = center
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def _sat_int(v):
    """saturate_cast<int>(double): round half to even (cvRound / lrint), clamped to int32."""
    import numpy as np

    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_affine_coeffs(M):
    """The inverse map cv::warpAffine derives from the forward 2x3 matrix (imgwarp.cpp): dst(x, y) = src(A @ (x, y, 1))."""
    import numpy as np

    m = np.array(M, dtype=np.float64).reshape(2, 3).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0] = A11
    m[0, 1] *= -D
    m[1, 0] *= -D
    m[1, 1] = A22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    return m


def warp_affine_fixed_point(A, width, height):
    """Fixed-point source coordinates of every destination pixel exactly as WarpAffineInvoker forms them: AB_BITS = 10,
    INTER_BITS = 5, round_delta = AB_SCALE / INTER_TAB_SIZE / 2.  Returns int64 (X, Y) arrays (height, width) with 5 fractional bits."""
    import numpy as np

    AB, IB = 10, 5
    x = np.arange(width, dtype=np.float64)
    y = np.arange(height, dtype=np.float64)
    adelta = _sat_int(A[0, 0] * x * (1 << AB))
    bdelta = _sat_int(A[1, 0] * x * (1 << AB))
    rd = (1 << AB) // (1 << IB) // 2
    X0 = _sat_int((A[0, 1] * y + A[0, 2]) * (1 << AB)) + rd
    Y0 = _sat_int((A[1, 1] * y + A[1, 2]) * (1 << AB)) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB - IB)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB - IB)
    return X, Y


def cv2_warp_affine(src, M, dsize, dst=None, flags=1, borderMode=0, borderValue=(0, 0, 0)):
    """cv2.warpAffine(im, M[:2], dsize=(w, h), borderValue=(114, 114, 114)) for 8-bit images, INTER_LINEAR + BORDER_CONSTANT;
    call site utils/augmentations.py:166.  Each destination pixel: fixed-point source position (above), integer part (sx, sy),
    5-bit fractions (fx, fy); the four neighbours (constant border value where outside the image) weighted by
    (32 - fx)(32 - fy), fx (32 - fy), (32 - fx) fy, fx fy (x 32 = 15-bit weights that sum to 32768); out = (sum + 2^14) >> 15."""
    import numpy as np

    assert flags == 1 and borderMode == 0 and src.dtype == np.uint8
    s3 = src[:, :, None] if src.ndim == 2 else src
    sh, sw, cn = s3.shape
    w, h = int(dsize[0]), int(dsize[1])
    A = warp_affine_coeffs(M)
    X, Y = warp_affine_fixed_point(A, w, h)
    sx = np.clip(X >> 5, -32768, 32767)  # saturate_cast<short>
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    bv = np.zeros(cn, dtype=np.int64)
    bvs = np.atleast_1d(np.asarray(borderValue, dtype=np.float64))
    for c in range(cn):
        bv[c] = int(np.clip(np.rint(bvs[c] if c < len(bvs) else 0.0), 0, 255))
    a = s3.astype(np.int64)

    def sample(yy, xx):
        inside = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        v = a[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return np.where(inside[..., None], v, bv[None, None, :])

    w00 = ((32 - fx) * (32 - fy) * 32)[..., None]
    w01 = (fx * (32 - fy) * 32)[..., None]
    w10 = ((32 - fx) * fy * 32)[..., None]
    w11 = (fx * fy * 32)[..., None]
    out = (sample(sy, sx) * w00 + sample(sy, sx + 1) * w01 + sample(sy + 1, sx) * w10 + sample(sy + 1, sx + 1) * w11 + (1 << 14)) >> 15
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if src.ndim == 2 else out


def _hsv_div_tables():
    import numpy as np

    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, dtype=np.int64)
    hdiv = np.zeros(256, dtype=np.int64)
    sdiv[1:] = _sat_int((255 << 12) / (1.0 * i))
    hdiv[1:] = _sat_int((180 << 12) / (6.0 * i))
    return sdiv, hdiv


def cv2_bgr2hsv(im):
    """cv2.cvtColor(im, cv2.COLOR_BGR2HSV) for uint8 (H in 0..179): RGB2HSV_b's integer arithmetic, hsv_shift = 12."""
    import numpy as np

    sdiv, hdiv = _hsv_div_tables()
    b, g, r = (im[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr, vg = v == r, v == g
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack((np.clip(h, 0, 255), s, v), -1).astype(np.uint8)


def cv2_hsv2bgr(hsv):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) for uint8: HSV2RGB_b = the float conversion on (h, s / 255, v / 255) with hscale = 6 / 180,
    result * 255 rounded (half to even) and saturated."""
    import numpy as np

    f32 = np.float32
    h = hsv[..., 0].astype(f32) * f32(6.0 / 180.0)
    s = hsv[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = hsv[..., 2].astype(f32) * f32(1.0 / 255.0)
    h = np.where(h < 0, h + f32(6), h)
    h = np.where(h >= 6, h - f32(6), h)
    sector = np.floor(h).astype(np.int64)
    frac = (h - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    frac = np.where(bad, f32(0), frac)
    one = f32(1.0)
    tab = np.stack((v, (v * (one - s)).astype(f32), (v * (one - (s * frac).astype(f32))).astype(f32),
                    (v * (one - (s * (one - frac)).astype(f32))).astype(f32)), -1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]], dtype=np.int64)  # (b, g, r) <- tab index per sector
    idx = sd[sector]
    bgr = np.take_along_axis(tab, idx, -1)
    grey = (hsv[..., 1] == 0)[..., None]
    bgr = np.where(grey, v[..., None], bgr)
    out = np.rint((bgr * f32(255.0)).astype(f32))
    return np.clip(out, 0, 255).astype(np.uint8)


def cv2_cvt_color(src, code, dst=None):
    """cv2.cvtColor for the two codes augment_hsv uses (utils/augmentations.py:73,83); `dst=` is filled in place like OpenCV does."""
    out = cv2_bgr2hsv(src) if code == COLOR_BGR2HSV else cv2_hsv2bgr(src) if code == COLOR_HSV2BGR else None
    if out is None:
        raise NotImplementedError(f"cvtColor code {code}")
    if dst is not None:
        dst[...] = out
        return dst
    return out


def cv2_lut(src, lut):
    return lut[src]


def cv2_split(m):
    return [m[..., k] for k in range(m.shape[-1])]


def cv2_merge(mv):
    import numpy as np

    return np.stack(list(mv), -1)


def xywhn2xyxy(x, w=640, h=640, padw=0, padh=0):
    """ultralytics.utils.ops.xywhn2xyxy; call sites utils/dataloaders.py:721,838: normalised (cx, cy, w, h) -> pixel corners + padding."""
    y = x.clone() if isinstance(x, torch.Tensor) else x.copy()
    y[..., 0] = w * (x[..., 0] - x[..., 2] / 2) + padw
    y[..., 1] = h * (x[..., 1] - x[..., 3] / 2) + padh
    y[..., 2] = w * (x[..., 0] + x[..., 2] / 2) + padw
    y[..., 3] = h * (x[..., 1] + x[..., 3] / 2) + padh
    return y


def xyxy2xywhn(x, w=640, h=640, clip=False, eps=0.0):
    """ultralytics.utils.ops.xyxy2xywhn; call site utils/dataloaders.py:737: pixel corners -> normalised (cx, cy, w, h), optionally
    clipped to the image (minus eps) IN PLACE first."""
    if clip:
        clip_boxes(x, (h - eps, w - eps))
    y = x.clone() if isinstance(x, torch.Tensor) else x.copy()
    y[..., 0] = ((x[..., 0] + x[..., 2]) / 2) / w
    y[..., 1] = ((x[..., 1] + x[..., 3]) / 2) / h
    y[..., 2] = (x[..., 2] - x[..., 0]) / w
    y[..., 3] = (x[..., 3] - x[..., 1]) / h
    return y
