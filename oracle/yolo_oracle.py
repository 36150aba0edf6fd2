"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (yolov5_amd/).

CPU restatement (torch-CPU fp32 for the floating-point layers, numpy for index/selection work) of
the reference hot path named by BASELINE.json `north_star`:

    forward (Conv/C3/SPPF/Upsample/Concat/Detect/Segment)  -> models/common.py, models/yolo.py
    non_max_suppression                                     -> utils/general.py:658-767
    ComputeLoss (+build_targets)                            -> utils/loss.py:101-247
    process_mask / crop_mask                                -> utils/segment/general.py:10-51
    scale_boxes                                             -> utils/general.py:613-626
    process_batch / ap_per_class (validation metrics)       -> utils/metrics.py:25-126,224-265, val.py:296-307
    fuse_conv_and_bn                                        -> utils/torch_utils.py:224-254

Every function cites the reference file:line it follows.  The restatement is PINNED two ways:
(1) `tests/test_oracle_vs_reference.py` runs it against the unmodified reference modules imported through
`oracle/ref_shim.py` (only where /root/reference exists), and (2) `tests/test_oracle_golden.py` checks it
against fixtures under tests/golden/ that `oracle/make_golden.py` produced from that same imported
reference.  The third-party arithmetic underneath the reference (`ultralytics` bbox_iou/xywh2xyxy/..,
`torchvision.ops.nms`) is absent from /root/reference and from this image -> that layer is
"PARITY UNPINNED" (see oracle/thirdparty.py); its restatement follows the published upstream algorithm.

Tie / order contracts (SURVEY 8c hazards): NMS sort is stable (equal scores keep the lower row first);
`tobj[b,a,gj,gi] = iou` with duplicate indices is last-write-wins in build_targets row order.
"""
from __future__ import annotations

import math

import re

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------
# model configs (models/yolov5{n,s,m,l,x}.yaml, models/segment/yolov5*-seg.yaml)
# --------------------------------------------------------------------------------------------------
_MULT = {"n": (0.33, 0.25), "s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}
_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def model_cfg(name: str = "yolov5s", nc: int = 80) -> dict:
    """Dict equivalent of models/yolov5s.yaml:9-53 (and the -seg variants, models/segment/yolov5s-seg.yaml)."""
    seg = name.endswith("-seg")
    size = name.replace("-seg", "")[-1]
    gd, gw = _MULT[size]
    backbone = [
        [-1, 1, "Conv", [64, 6, 2, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C3", [128]],
        [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C3", [256]], [-1, 1, "Conv", [512, 3, 2]],
        [-1, 9, "C3", [512]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 3, "C3", [1024]],
        [-1, 1, "SPPF", [1024, 5]],
    ]
    head = [
        [-1, 1, "Conv", [512, 1, 1]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]],
        [[-1, 6], 1, "Concat", [1]], [-1, 3, "C3", [512, False]],
        [-1, 1, "Conv", [256, 1, 1]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]],
        [[-1, 4], 1, "Concat", [1]], [-1, 3, "C3", [256, False]],
        [-1, 1, "Conv", [256, 3, 2]], [[-1, 14], 1, "Concat", [1]], [-1, 3, "C3", [512, False]],
        [-1, 1, "Conv", [512, 3, 2]], [[-1, 10], 1, "Concat", [1]], [-1, 3, "C3", [1024, False]],
        [[17, 20, 23], 1, "Segment" if seg else "Detect",
         ["nc", "anchors", 32, 256] if seg else ["nc", "anchors"]],
    ]
    return {"nc": nc, "depth_multiple": gd, "width_multiple": gw, "anchors": [list(a) for a in _ANCHORS],
            "backbone": backbone, "head": head}


def _make_divisible(x, d):
    return math.ceil(x / d) * d


def parse_graph(cfg: dict, ch: int = 3):
    """Layer list with resolved channels/repeats; follows models/yolo.py:375-458 (parse_model)."""
    nc, gd, gw = cfg["nc"], cfg["depth_multiple"], cfg["width_multiple"]
    anchors = cfg["anchors"]
    na = len(anchors[0]) // 2
    no = na * (nc + 5)
    chs, layers, save = [ch], [], []
    c2 = ch
    for i, (f, n, m, args) in enumerate(cfg["backbone"] + cfg["head"]):
        args = [nc if a == "nc" else anchors if a == "anchors" else a for a in args]
        n = max(round(n * gd), 1) if n > 1 else n  # yolo.py:401
        spec = {"i": i, "f": f, "type": m}
        if m in ("Conv", "C3", "SPPF"):
            c1, c2 = chs[f], args[0]
            if c2 != no:
                c2 = _make_divisible(c2 * gw, 8)  # yolo.py:422-423
            if m == "Conv":
                k = args[1] if len(args) > 1 else 1
                s = args[2] if len(args) > 2 else 1
                p = args[3] if len(args) > 3 else None
                spec.update(c1=c1, c2=c2, k=k, s=s, p=k // 2 if p is None else p)
            elif m == "C3":
                spec.update(c1=c1, c2=c2, n=n, shortcut=args[1] if len(args) > 1 else True)
            else:
                spec.update(c1=c1, c2=c2, k=args[1] if len(args) > 1 else 5)
        elif m == "Concat":
            c2 = sum(chs[x] for x in f)
        elif m in ("Detect", "Segment"):
            spec.update(nc=args[0], anchors=args[1], ch=[chs[x] for x in f])
            if m == "Segment":
                spec.update(nm=args[2], npr=_make_divisible(args[3] * gw, 8))  # yolo.py:440-441
        else:  # nn.Upsample
            c2 = chs[f]
        spec["c_out"] = c2
        layers.append(spec)
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        if i == 0:
            chs = []
        chs.append(c2)
    return layers, sorted(save)


# --------------------------------------------------------------------------------------------------
# layers (models/common.py)
# --------------------------------------------------------------------------------------------------
BN_EPS = 1e-3  # set by ultralytics initialize_weights (models/yolo.py:259) -- restated, see thirdparty.py


def fuse_conv_and_bn(w, gamma, beta, mean, var, eps=BN_EPS, conv_bias=None):
    """utils/torch_utils.py:224-254: W' = diag(g/sqrt(eps+var)) W ; b' = W_bn b_conv + beta - g*mean/sqrt(var+eps)."""
    co = w.shape[0]
    w_bn = torch.diag(gamma.div(torch.sqrt(eps + var)))
    wf = torch.mm(w_bn, w.reshape(co, -1)).view(w.shape)
    b_conv = torch.zeros(co, dtype=w.dtype) if conv_bias is None else conv_bias
    b_bn = beta - gamma.mul(mean).div(torch.sqrt(var + eps))
    bf = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn
    return wf, bf


BN_MOMENTUM = 0.03
_BN_TRAIN = [False]  # set by model_forward(training=True, bn_batch_stats=True)


def _conv(sd, p, x, k, s, pad, act=True):
    """models/common.py:74-92 Conv.forward (conv->BN(eval)->SiLU) or forward_fuse (conv+bias->SiLU)."""
    w = sd[p + ".conv.weight"]
    if (p + ".bn.weight") in sd:
        y = F.conv2d(x, w, None, s, pad)
        # train.py runs the model in train() mode: BatchNorm2d uses batch statistics and updates the running ones in
        # place (momentum 0.03 / eps 1e-3 from initialize_weights, models/yolo.py:259)
        y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                         sd[p + ".bn.bias"], _BN_TRAIN[0], BN_MOMENTUM if _BN_TRAIN[0] else 0.0, BN_EPS)
    else:
        y = F.conv2d(x, w, sd[p + ".conv.bias"], s, pad)
    return F.silu(y) if act else y


def _bottleneck(sd, p, x, shortcut):
    """models/common.py:164-181 (e=1.0 inside C3, common.py:242): x + cv2(cv1(x)) iff shortcut and c1==c2."""
    y = _conv(sd, p + ".cv2", _conv(sd, p + ".cv1", x, 1, 1, 0), 3, 1, 1)
    return x + y if shortcut else y


def _c3(sd, p, x, n, shortcut):
    """models/common.py:230-246: cv3(cat(m(cv1(x)), cv2(x)))."""
    a = _conv(sd, p + ".cv1", x, 1, 1, 0)
    for j in range(n):
        a = _bottleneck(sd, f"{p}.m.{j}", a, shortcut)
    b = _conv(sd, p + ".cv2", x, 1, 1, 0)
    return _conv(sd, p + ".cv3", torch.cat((a, b), 1), 1, 1, 0)


def _sppf(sd, p, x, k):
    """models/common.py:318-340: cv2(cat(x, m(x), m(m(x)), m(m(m(x))))) with MaxPool2d(k,1,k//2)."""
    x = _conv(sd, p + ".cv1", x, 1, 1, 0)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return _conv(sd, p + ".cv2", torch.cat((x, y1, y2, y3), 1), 1, 1, 0)


def _proto(sd, p, x):
    """models/common.py:1104-1117: cv3(cv2(upsample2x(cv1(x))))."""
    y = _conv(sd, p + ".cv1", x, 3, 1, 1)
    y = F.interpolate(y, scale_factor=2, mode="nearest")
    return _conv(sd, p + ".cv3", _conv(sd, p + ".cv2", y, 3, 1, 1), 1, 1, 0)


def make_grid(nx, ny, anchors_i, stride_i, na=3):
    """models/yolo.py:117-128: grid = (ix-0.5, iy-0.5); anchor_grid = anchors(grid units)*stride."""
    yv, xv = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32), indexing="ij")
    grid = torch.stack((xv, yv), 2).expand(1, na, ny, nx, 2) - 0.5
    anchor_grid = (anchors_i * stride_i).view(1, na, 1, 1, 2).expand(1, na, ny, nx, 2)
    return grid, anchor_grid


def detect_head(sd, p, xs, anchors, strides, nc, nm=0, training=False):
    """models/yolo.py:91-115 Detect.forward. anchors: (nl,na,2) in GRID units (after yolo.py:254)."""
    na = anchors.shape[1]
    no = nc + 5 + nm
    z, raw = [], []
    for i, x in enumerate(xs):
        x = F.conv2d(x, sd[f"{p}.m.{i}.weight"], sd[f"{p}.m.{i}.bias"])
        bs, _, ny, nx = x.shape
        x = x.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        raw.append(x)
        if not training:
            grid, ag = make_grid(nx, ny, anchors[i], strides[i], na)
            if nm:  # Segment: mask coefficients are NOT sigmoided (yolo.py:104-108)
                xy, wh, conf, mask = x.split((2, 2, nc + 1, nm), 4)
                xy = (xy.sigmoid() * 2 + grid) * strides[i]
                wh = (wh.sigmoid() * 2) ** 2 * ag
                y = torch.cat((xy, wh, conf.sigmoid(), mask), 4)
            else:
                xy, wh, conf = x.sigmoid().split((2, 2, nc + 1), 4)
                xy = (xy * 2 + grid) * strides[i]
                wh = (wh * 2) ** 2 * ag
                y = torch.cat((xy, wh, conf), 4)
            z.append(y.view(bs, na * nx * ny, no))
    return raw if training else (torch.cat(z, 1), raw)


def model_strides(cfg):
    """models/yolo.py:250-256 derives strides from a 256^2 probe; for these graphs it is 8/16/32 by construction."""
    layers, _ = parse_graph(cfg)
    det = layers[-1]
    red = []
    scale = {}
    for L in layers[:-1]:
        f = L["f"]
        src = scale.get(L["i"] - 1, 1) if f == -1 else None
        if L["type"] == "Conv":
            scale[L["i"]] = (scale.get(L["i"] - 1, 1) if f == -1 else scale[f]) * L["s"]
        elif L["type"] == "nn.Upsample":
            scale[L["i"]] = scale[L["i"] - 1] // 2
        elif L["type"] == "Concat":
            scale[L["i"]] = scale[L["i"] - 1] if f[0] == -1 else scale[f[0]]
        else:
            scale[L["i"]] = src if f == -1 else scale[f]
    for j in det["f"]:
        red.append(float(scale[j]))
    return torch.tensor(red)


def model_anchors(cfg):
    """Anchors in grid units: yaml pixels / stride (models/yolo.py:254)."""
    a = torch.tensor(cfg["anchors"], dtype=torch.float32).view(len(cfg["anchors"]), -1, 2)
    return a / model_strides(cfg).view(-1, 1, 1)


def model_forward(cfg, sd, x, training=False, bn_batch_stats=False):
    """models/yolo.py:160-170 `_forward_once` over the parsed graph; returns what the reference returns:
    training -> list of raw (bs,na,ny,nx,no); eval -> (z, raw) (+ proto for Segment: (z, proto, raw)).
    bn_batch_stats=True reproduces model.train(): every BatchNorm uses (and updates) batch statistics."""
    _BN_TRAIN[0] = bool(bn_batch_stats)
    try:
        return _model_forward(cfg, sd, x, training)
    finally:
        _BN_TRAIN[0] = False


def scale_img(img, ratio=1.0, same_shape=False, gs=32):
    """utils/torch_utils.py scale_img."""
    if ratio == 1.0:
        return img
    h, w = img.shape[2:]
    s = (int(h * ratio), int(w * ratio))
    img = F.interpolate(img, size=s, mode="bilinear", align_corners=False)
    if not same_shape:
        h, w = (math.ceil(v * ratio / gs) * gs for v in (h, w))
    return F.pad(img, [0, w - s[1], 0, h - s[0]], value=0.447)


def forward_augment(cfg, sd, x):
    """models/yolo.py:269-312 `_forward_augment` + `_descale_pred` + `_clip_augmented` (eval mode): returns the concatenated z."""
    img_size = x.shape[-2:]
    gs = int(max(model_strides(cfg)))
    y = []
    for si, fi in zip((1, 0.83, 0.67), (None, 3, None)):
        xi = scale_img(x.flip(fi) if fi else x, si, gs=gs)
        p = model_forward(cfg, sd, xi)[0].clone()
        p[..., :4] /= si
        if fi == 2:
            p[..., 1] = img_size[0] - p[..., 1]
        elif fi == 3:
            p[..., 0] = img_size[1] - p[..., 0]
        y.append(p)
    nl = len(model_strides(cfg))
    g = sum(4 ** k for k in range(nl))
    y[0] = y[0][:, :-(y[0].shape[1] // g)]
    y[-1] = y[-1][:, (y[-1].shape[1] // g) * 4 ** (nl - 1):]
    return torch.cat(y, 1)


def _model_forward(cfg, sd, x, training=False):
    layers, save = parse_graph(cfg, x.shape[1])
    y = []
    strides = model_strides(cfg)
    anchors = sd.get(f"model.{len(layers) - 1}.anchors", None)
    if anchors is None:
        anchors = model_anchors(cfg)
    for L in layers:
        f, i, t = L["f"], L["i"], L["type"]
        if f != -1:
            x = y[f] if isinstance(f, int) else [x if j == -1 else y[j] for j in f]
        p = f"model.{i}"
        if t == "Conv":
            x = _conv(sd, p, x, L["k"], L["s"], L["p"])
        elif t == "C3":
            x = _c3(sd, p, x, L["n"], L["shortcut"])
        elif t == "SPPF":
            x = _sppf(sd, p, x, L["k"])
        elif t == "nn.Upsample":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif t == "Concat":
            x = torch.cat(x, 1)
        elif t == "Detect":
            x = detect_head(sd, p, list(x), anchors, strides, L["nc"], 0, training)
        elif t == "Segment":
            proto = _proto(sd, p + ".proto", x[0])
            d = detect_head(sd, p, list(x), anchors, strides, L["nc"], L["nm"], training)
            x = (d, proto) if training else (d[0], proto, d[1])  # yolo.py:148-150
        y.append(x if i in save else None)
    return x


# --------------------------------------------------------------------------------------------------
# NMS (utils/general.py:658-767) -- numpy, dtype-preserving
# --------------------------------------------------------------------------------------------------
def greedy_nms(boxes: np.ndarray, scores: np.ndarray, iou_thres: float) -> np.ndarray:
    """torchvision.ops.nms semantics (call site general.py:750); see oracle/thirdparty.py:nms for the contract."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    b = boxes[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    sup = np.zeros(n, dtype=bool)
    keep = []
    zero, thr = b.dtype.type(0), b.dtype.type(iou_thres)
    for i in range(n):
        if sup[i]:
            continue
        keep.append(order[i])
        if i + 1 < n:
            w = np.maximum(zero, np.minimum(x2[i], x2[i + 1:]) - np.maximum(x1[i], x1[i + 1:]))
            h = np.maximum(zero, np.minimum(y2[i], y2[i + 1:]) - np.maximum(y1[i], y1[i + 1:]))
            inter = w * h
            with np.errstate(divide="ignore", invalid="ignore"):
                sup[i + 1:] |= inter / (areas[i] + areas[i + 1:] - inter) > thr
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, max_det=300, nm=0, max_nms=30000, max_wh=7680, labels=()):
    """utils/general.py:658-767 restated on numpy float32 (the reference's arithmetic dtype for fp32 input).

    prediction (bs, n, 5+nc+nm) -> list of (k, 6+nm) float32 arrays [x1,y1,x2,y2,conf,cls,(mask..)].
    The wall-clock `time_limit` break (general.py:692,763-765) is not restated (time dependent).
    """
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1  # general.py:675-676
    pred = np.asarray(prediction, dtype=np.float32)
    bs = pred.shape[0]
    nc = pred.shape[2] - nm - 5
    mi = 5 + nc
    multi_label = multi_label and nc > 1  # general.py:693
    conf_t = np.float32(conf_thres)
    out = [np.zeros((0, 6 + nm), dtype=np.float32) for _ in range(bs)]
    for xi in range(bs):
        x = pred[xi]
        x = x[x[:, 4] > conf_t].copy()  # general.py:679,703
        if len(labels) and len(labels[xi]):  # general.py:706-712: a-priori labels (autolabelling) join the candidates with confidence 1
            lb = np.asarray(labels[xi], dtype=np.float32)
            v = np.zeros((len(lb), nc + nm + 5), dtype=np.float32)
            v[:, :4] = lb[:, 1:5]
            v[:, 4] = 1.0
            v[np.arange(len(lb)), lb[:, 0].astype(np.int64) + 5] = 1.0
            x = np.concatenate((x, v), 0)
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]  # general.py:719  (note: also scales the mask columns, as the reference does)
        half = x[:, 2:4] / np.float32(2)
        box = np.concatenate((x[:, :2] - half, x[:, :2] + half), 1)  # xywh2xyxy, general.py:722
        mask = x[:, mi:]
        if multi_label:  # general.py:726-728
            i, j = np.nonzero(x[:, 5:mi] > conf_t)
            x = np.concatenate((box[i], x[i, 5 + j, None], j[:, None].astype(np.float32), mask[i]), 1)
        else:  # general.py:729-731 (max returns the FIRST maximal index)
            j = x[:, 5:mi].argmax(1)
            conf = x[np.arange(x.shape[0]), 5 + j]
            x = np.concatenate((box, conf[:, None], j[:, None].astype(np.float32), mask), 1)[conf > conf_t]
        if classes is not None:  # general.py:734-735
            x = x[np.isin(x[:, 5], np.asarray(classes, dtype=np.float32))]
        n = x.shape[0]
        if not n:
            continue
        x = x[np.argsort(-x[:, 4].astype(np.float64), kind="stable")[:max_nms]]  # general.py:745 (stable contract)
        c = x[:, 5:6] * np.float32(0 if agnostic else max_wh)  # general.py:748
        keep = greedy_nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]  # general.py:749-751
        out[xi] = x[keep]
    return out


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """utils/general.py:613-626 (+ clip_boxes): de-letterbox xyxy boxes in place (float32 numpy)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    boxes[..., [0, 2]] -= np.float32(pad[0])
    boxes[..., [1, 3]] -= np.float32(pad[1])
    boxes[..., :4] /= np.float32(gain)
    boxes[..., [0, 2]] = boxes[..., [0, 2]].clip(0, img0_shape[1])
    boxes[..., [1, 3]] = boxes[..., [1, 3]].clip(0, img0_shape[0])
    return boxes


# --------------------------------------------------------------------------------------------------
# validation metrics (utils/metrics.py:25-126,224-265; val.py:296-307)
# --------------------------------------------------------------------------------------------------
def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    """Restatement of utils/augmentations.py:85-115 on the restated OpenCV primitives (oracle/thirdparty.py: cv2_resize,
    cv2_copy_make_border): uint8 HWC image -> (letterboxed image, (rw, rh), (dw, dh)).  Pinned against the reference's own
    function through tests/golden/letterbox.npz (tests/test_oracle_golden.py)."""
    from . import thirdparty as tp

    h0, w0 = im.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / h0, new_shape[1] / w0)                      # :92
    if not scaleup:
        r = min(r, 1.0)                                                # :93-94
    ratio = (r, r)
    new_unpad = (round(w0 * r), round(h0 * r))                         # :98 (w, h)
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]  # :99
    if auto:
        dw, dh = dw % stride, dh % stride                              # :101 minimum rectangle
    elif scaleFill:
        dw, dh = 0.0, 0.0                                              # :103-105 stretch
        new_unpad = (new_shape[1], new_shape[0])
        ratio = (new_shape[1] / w0, new_shape[0] / h0)
    dw, dh = dw / 2, dh / 2                                            # :107-108
    if (w0, h0) != new_unpad:
        im = tp.cv2_resize(im, new_unpad, interpolation=1)             # :110-111 INTER_LINEAR
    top, bottom = round(dh - 0.1), round(dh + 0.1)                     # :112
    left, right = round(dw - 0.1), round(dw + 0.1)                     # :113
    im = tp.cv2_copy_make_border(im, top, bottom, left, right, 0, value=color)  # :114 BORDER_CONSTANT
    return im, ratio, (dw, dh)


def box_iou_np(box1, box2, eps=1e-7):
    """ultralytics.utils.metrics.box_iou (call site utils/metrics.py:252), float32 numpy, same operation order."""
    box1 = np.asarray(box1, np.float32)
    box2 = np.asarray(box2, np.float32)
    a1, a2 = box1[:, None, :2], box1[:, None, 2:4]
    b1, b2 = box2[None, :, :2], box2[None, :, 2:4]
    wh = np.clip(np.minimum(a2, b2) - np.maximum(a1, b1), 0, None).astype(np.float32)
    inter = wh[..., 0] * wh[..., 1]
    area1 = (a2 - a1)[..., 0] * (a2 - a1)[..., 1]
    area2 = (b2 - b1)[..., 0] * (b2 - b1)[..., 1]
    return inter / (area1 + area2 - inter + np.float32(eps))


def process_batch(detections, labels, iouv):
    """utils/metrics.py:224-265 (box branch): (N,6) [x1,y1,x2,y2,conf,cls] x (M,5) [cls,x1,y1,x2,y2] -> bool (N, niou).
    The sort at :260 is made STABLE (the reference's numpy argsort()[::-1] is stable only for short lists): equal IoUs keep
    the later (label-major, detection-minor) pair first after the reversal."""
    detections = np.asarray(detections, np.float32)
    labels = np.asarray(labels, np.float32)
    iouv = np.asarray(iouv, np.float32)
    correct = np.zeros((detections.shape[0], iouv.shape[0]), dtype=bool)
    if not detections.shape[0] or not labels.shape[0]:
        return correct
    iou = box_iou_np(labels[:, 1:], detections[:, :4])  # :252
    correct_class = labels[:, 0:1] == detections[:, 5]  # :255
    for i in range(len(iouv)):
        li, di = np.nonzero((iou >= iouv[i]) & correct_class)  # :257
        if li.shape[0]:
            matches = np.stack([li.astype(np.float64), di.astype(np.float64), iou[li, di].astype(np.float64)], 1)  # :259
            if li.shape[0] > 1:
                matches = matches[np.argsort(matches[:, 2], kind="stable")[::-1]]  # :260
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]  # :261
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]  # :263
            correct[matches[:, 1].astype(int), i] = True  # :264
    return correct


def val_match_image(pred, targets_px, im_shape, shape0, ratio_pad, iouv):
    """val.py:296-307 for one image: pred (n,6+) NMS rows, targets_px (m,5) [cls, cx, cy, w, h] in letterboxed pixels
    (val.py:274) -> (correct (n,niou) bool, predn boxes (n,4))."""
    predn = np.array(pred[:, :6], np.float32, copy=True)
    scale_boxes(im_shape, predn[:, :4], shape0, ratio_pad)  # :298
    t = np.asarray(targets_px, np.float32)
    half = t[:, 3:5] / np.float32(2)
    tbox = np.concatenate([t[:, 1:3] - half, t[:, 1:3] + half], 1).astype(np.float32)  # xywh2xyxy, :303
    scale_boxes(im_shape, tbox, shape0, ratio_pad)  # :304
    labelsn = np.concatenate([t[:, 0:1], tbox], 1)  # :305
    return process_batch(predn, labelsn, iouv), predn[:, :4]


def compute_ap(recall, precision):
    """utils/metrics.py:98-126, method 'interp': 101-point interpolated area under the precision envelope."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))  # :113
    x = np.linspace(0, 1, 101)  # :118
    y = np.interp(x, mrec, mpre)
    ap = float(np.sum((y[1:] + y[:-1]) * np.diff(x) / 2.0))  # np.trapz / np.trapezoid, :119
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """utils/metrics.py:25-95 without the plots -> tp, fp, p, r, f1, ap (nc, niou), unique classes."""
    from .thirdparty import smooth

    order = np.argsort(-conf)  # :46
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    unique_classes, nt = np.unique(target_cls, return_counts=True)  # :50
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        sel = pred_cls == c
        n_l, n_p = nt[ci], sel.sum()
        if n_p == 0 or n_l == 0:  # :60
            continue
        fpc = (1 - tp[sel]).cumsum(0)  # :64
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)  # :68
        r[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)  # :69
        precision = tpc / (tpc + fpc)  # :72
        p[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)  # :73
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]  # :77
    f1 = 2 * p * r / (p + r + eps)  # :82
    i = smooth(f1.mean(0), 0.1).argmax()  # :91
    p, r, f1 = p[:, i], r[:, i], f1[:, i]
    tpn = (r * nt).round()  # :93
    fpn = (tpn / (p + eps) - tpn).round()  # :94
    return tpn, fpn, p, r, f1, ap, unique_classes.astype(int)


# --------------------------------------------------------------------------------------------------
# ComputeLoss (utils/loss.py:101-247)
# --------------------------------------------------------------------------------------------------
HYP_SCRATCH_LOW = {"box": 0.05, "cls": 0.5, "cls_pw": 1.0, "obj": 1.0, "obj_pw": 1.0, "anchor_t": 4.0,
                   "fl_gamma": 0.0, "label_smoothing": 0.0}  # data/hyps/hyp.scratch-low.yaml:17-24


def build_targets(shapes, targets, anchors, anchor_t=4.0):
    """utils/loss.py:185-247.  shapes: list of p[i].shape (bs,na,ny,nx,no); targets (nt,6) float32 torch;
    anchors (nl,na,2) grid units.  Returns tcls, tbox, indices, anch exactly as the reference."""
    na, nt = anchors.shape[1], targets.shape[0]
    tcls, tbox, indices, anch = [], [], [], []
    gain = torch.ones(7)
    ai = torch.arange(na).float().view(na, 1).repeat(1, nt)
    targets = torch.cat((targets.repeat(na, 1, 1), ai[..., None]), 2)
    g = 0.5
    off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]]).float() * g
    for i in range(len(shapes)):
        a_i, shape = anchors[i], shapes[i]
        gain[2:6] = torch.tensor(shape)[[3, 2, 3, 2]]
        t = targets * gain
        if nt:
            r = t[..., 4:6] / a_i[:, None]
            j = torch.max(r, 1 / r).max(2)[0] < anchor_t
            t = t[j]
            gxy = t[:, 2:4]
            gxi = gain[[2, 3]] - gxy
            j, k = ((gxy % 1 < g) & (gxy > 1)).T
            l, m = ((gxi % 1 < g) & (gxi > 1)).T
            j = torch.stack((torch.ones_like(j), j, k, l, m))
            t = t.repeat((5, 1, 1))[j]
            offsets = (torch.zeros_like(gxy)[None] + off[:, None])[j]
        else:
            t = targets[0]
            offsets = 0
        bc, gxy, gwh, a = t.chunk(4, 1)
        a, (b, c) = a.long().view(-1), bc.long().T
        gij = (gxy - offsets).long()
        gi, gj = gij.T
        indices.append((b, a, gj.clamp_(0, shape[2] - 1), gi.clamp_(0, shape[3] - 1)))
        tbox.append(torch.cat((gxy - gij, gwh), 1))
        anch.append(a_i[a])
        tcls.append(c)
    return tcls, tbox, indices, anch


def bbox_ciou(box1, box2, eps=1e-7):
    """ultralytics bbox_iou(xywh=True, CIoU=True) as called at utils/loss.py:153 (see thirdparty.py)."""
    (x1, y1, w1, h1), (x2, y2, w2, h2) = box1.chunk(4, -1), box2.chunk(4, -1)
    b1_x1, b1_x2, b1_y1, b1_y2 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
    b2_x1, b2_x2, b2_y1, b2_y2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    inter = (b1_x2.minimum(b2_x2) - b1_x1.maximum(b2_x1)).clamp(0) * (b1_y2.minimum(b2_y2) - b1_y1.maximum(b2_y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = b1_x2.maximum(b2_x2) - b1_x1.minimum(b2_x1)
    ch = b1_y2.maximum(b2_y2) - b1_y1.minimum(b2_y1)
    c2 = cw.pow(2) + ch.pow(2) + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2).pow(2) + (b2_y1 + b2_y2 - b1_y1 - b1_y2).pow(2)) / 4
    v = (4 / math.pi ** 2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def compute_loss(p, targets, anchors, hyp=None, nc=80, balance=(4.0, 1.0, 0.4), autobalance_ssi=None):
    """utils/loss.py:134-183 (gr=1, sort_obj_iou off; hyp['fl_gamma'] > 0: focal loss).

    p: list of (bs,na,ny,nx,no) tensors (may require grad); returns (loss[1], loss_items[3]).
    autobalance_ssi = index of the stride-16 level (loss.py:127): `balance` must then be a LIST and is updated in place as
    loss.py:173-177 does (EMA of 1/obji per level in Python doubles, then normalised by the stride-16 entry)."""
    hyp = hyp or HYP_SCRATCH_LOW
    cp, cn = 1.0 - 0.5 * hyp.get("label_smoothing", 0.0), 0.5 * hyp.get("label_smoothing", 0.0)
    lcls, lbox, lobj = torch.zeros(1), torch.zeros(1), torch.zeros(1)
    tcls, tbox, indices, anch = build_targets([pi.shape for pi in p], targets, anchors, hyp["anchor_t"])
    pw_cls, pw_obj = torch.tensor([hyp["cls_pw"]]), torch.tensor([hyp["obj_pw"]])
    g = float(hyp.get("fl_gamma", 0.0))

    def bce(pred, true, pw):
        """BCEWithLogitsLoss(pos_weight), mean; fl_gamma > 0: wrapped in FocalLoss(gamma, alpha=0.25) -- utils/loss.py:77-98, :120-122."""
        if g <= 0:
            return F.binary_cross_entropy_with_logits(pred, true, pos_weight=pw)
        loss = F.binary_cross_entropy_with_logits(pred, true, pos_weight=pw, reduction="none")
        prob = pred.sigmoid()
        p_t = true * prob + (1 - true) * (1 - prob)
        return (loss * (true * 0.25 + (1 - true) * 0.75) * (1.0 - p_t) ** g).mean()

    for i, pi in enumerate(p):
        b, a, gj, gi = indices[i]
        tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype)
        n = b.shape[0]
        if n:
            pxy, pwh, _, pcls = pi[b, a, gj, gi].split((2, 2, 1, nc), 1)
            pxy = pxy.sigmoid() * 2 - 0.5
            pwh = (pwh.sigmoid() * 2) ** 2 * anch[i]
            pbox = torch.cat((pxy, pwh), 1)
            iou = bbox_ciou(pbox, tbox[i]).squeeze(-1)
            lbox = lbox + (1.0 - iou).mean()
            iou = iou.detach().clamp(0).type(tobj.dtype)
            # duplicate (b,a,gj,gi): CONTRACT = last write in build_targets row order wins (SURVEY 8c hazard 3).  torch's
            # CPU index_put_ gives exactly that for small n but an unspecified winner for a few thousand rows (observed),
            # and its CUDA kernel is unordered, so the contract is spelled out with numpy (documented: last value wins).
            lin = ((b * pi.shape[1] + a) * pi.shape[2] + gj) * pi.shape[3] + gi
            tobj.view(-1).numpy()[lin.numpy()] = iou.numpy()
            if nc > 1:
                t = torch.full_like(pcls, cn)
                t[range(n), tcls[i]] = cp
                lcls = lcls + bce(pcls, t, pw_cls)
        obji = bce(pi[..., 4], tobj, pw_obj)
        lobj = lobj + obji * balance[i]
        if autobalance_ssi is not None:
            balance[i] = balance[i] * 0.9999 + 0.0001 / obji.detach().item()  # loss.py:174
    if autobalance_ssi is not None:
        balance[:] = [x / balance[autobalance_ssi] for x in balance]  # loss.py:177
    lbox = lbox * hyp["box"]
    lobj = lobj * hyp["obj"]
    lcls = lcls * hyp["cls"]
    bs = p[0].shape[0]
    return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()


# --------------------------------------------------------------------------------------------------
# segmentation post-process (utils/segment/general.py:10-51)
# --------------------------------------------------------------------------------------------------
def crop_mask(masks, boxes):
    """utils/segment/general.py:10-22: zero everything outside box (r>=x1 & r<x2 & c>=y1 & c<y2)."""
    _n, h, w = masks.shape
    x1, y1, x2, y2 = torch.chunk(boxes[:, :, None], 4, 1)
    r = torch.arange(w, dtype=x1.dtype)[None, None, :]
    c = torch.arange(h, dtype=x1.dtype)[None, :, None]
    return masks * ((r >= x1) * (r < x2) * (c >= y1) * (c < y2))


def process_mask(protos, masks_in, bboxes, shape, upsample=False):
    """utils/segment/general.py:25-51: sigmoid(coef @ proto) -> crop to box/4 -> bilinear x4 -> >0.5."""
    c, mh, mw = protos.shape
    ih, iw = shape
    masks = (masks_in @ protos.float().view(c, -1)).sigmoid().view(-1, mh, mw)
    d = bboxes.clone()
    d[:, 0] *= mw / iw
    d[:, 2] *= mw / iw
    d[:, 3] *= mh / ih
    d[:, 1] *= mh / ih
    masks = crop_mask(masks, d)
    if upsample:
        masks = F.interpolate(masks[None], shape, mode="bilinear", align_corners=False)[0]
    return masks.gt_(0.5)


# --------------------------------------------------------------------------------------------------
# state-dict enumeration (names/shapes of the reference's unfused model) -- lets tests regenerate the
# golden weights with oracle/detgen.py without the reference being present
# --------------------------------------------------------------------------------------------------
class _Shape:
    def __init__(self, *shape):
        self.shape = tuple(shape)


def state_spec(cfg: dict, ch: int = 3) -> dict:
    """Ordered {name: _Shape} of `DetectionModel(cfg).state_dict()` (unfused: conv.weight + bn.*)."""
    layers, _ = parse_graph(cfg, ch)
    spec = {}

    def conv(p, c1, c2, k):
        spec[p + ".conv.weight"] = _Shape(c2, c1, k, k)
        spec[p + ".bn.weight"] = _Shape(c2)
        spec[p + ".bn.bias"] = _Shape(c2)
        spec[p + ".bn.running_mean"] = _Shape(c2)
        spec[p + ".bn.running_var"] = _Shape(c2)
        spec[p + ".bn.num_batches_tracked"] = _Shape()

    for L in layers:
        p, t = f"model.{L['i']}", L["type"]
        if t == "Conv":
            conv(p, L["c1"], L["c2"], L["k"])
        elif t == "C3":
            c_ = int(L["c2"] * 0.5)
            conv(p + ".cv1", L["c1"], c_, 1)
            conv(p + ".cv2", L["c1"], c_, 1)
            conv(p + ".cv3", 2 * c_, L["c2"], 1)
            for j in range(L["n"]):
                conv(f"{p}.m.{j}.cv1", c_, c_, 1)
                conv(f"{p}.m.{j}.cv2", c_, c_, 3)
        elif t == "SPPF":
            c_ = L["c1"] // 2
            conv(p + ".cv1", L["c1"], c_, 1)
            conv(p + ".cv2", c_ * 4, L["c2"], 1)
        elif t in ("Detect", "Segment"):
            nl = len(L["anchors"])
            na = len(L["anchors"][0]) // 2
            no = na * (L["nc"] + 5 + L.get("nm", 0))
            spec[p + ".anchors"] = _Shape(nl, na, 2)
            for j, c in enumerate(L["ch"]):
                spec[f"{p}.m.{j}.weight"] = _Shape(no, c, 1, 1)
                spec[f"{p}.m.{j}.bias"] = _Shape(no)
            if t == "Segment":
                conv(p + ".proto.cv1", L["ch"][0], L["npr"], 3)
                conv(p + ".proto.cv2", L["npr"], L["npr"], 3)
                conv(p + ".proto.cv3", L["npr"], L["nm"], 1)
    return spec


def det_state_dict(cfg: dict, seed: int = 0, fused: bool = False, bn_stats=None, head_affine=None, bn_gamma_scale=None) -> dict:
    """The deterministic weights the golden fixtures were generated with (torch fp32 tensors).  bn_stats / head_affine: the
    full-resolution detset_* fixtures (oracle/make_golden.py:gen_detset -> detgen.condition_state_dict)."""
    from . import detgen

    spec = state_spec(cfg)
    vals = detgen.fill_state_dict(spec, seed)
    if bn_stats is not None or head_affine is not None or bn_gamma_scale is not None:
        detgen.condition_state_dict(vals, bn_stats=bn_stats, head_affine=head_affine, bn_gamma_scale=bn_gamma_scale)
    sd = {}
    for k, v in vals.items():
        if v is None:
            sd[k] = model_anchors(cfg)
        else:
            sd[k] = torch.from_numpy(np.ascontiguousarray(v))
    if fused:
        out = {}
        for k in list(sd):
            if k.endswith(".conv.weight") and k.replace(".conv.weight", ".bn.weight") in sd:
                p = k[: -len(".conv.weight")]
                w, b = fuse_conv_and_bn(sd[k], sd[p + ".bn.weight"], sd[p + ".bn.bias"], sd[p + ".bn.running_mean"],
                                        sd[p + ".bn.running_var"])
                out[p + ".conv.weight"], out[p + ".conv.bias"] = w, b
            elif ".bn." not in k:
                out[k] = sd[k]
        sd = out
    return sd
