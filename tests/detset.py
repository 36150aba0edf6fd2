"""TEST HELPER: the full-resolution fixtures tests/golden/detset_*.npz (made by oracle/make_golden.py:gen_detset from the
unmodified reference at BASELINE configs C2 / C4 / C5) and the detection-set agreement measure used against them."""
import os

import numpy as np
import torch

from oracle import detgen, yolo_oracle as yo

G = os.path.join(os.path.dirname(__file__), "golden")

CASES = {  # fixture -> (model, image size, batch, seed, segmentation)
    "yolov5s_640": ("yolov5s", 640, 2, 3, False),
    "yolov5x_1280": ("yolov5x", 1280, 1, 4, False),
    "yolov5s-seg_640": ("yolov5s-seg", 640, 2, 5, True),
}


def load(name):
    g = np.load(os.path.join(G, f"detset_{name}.npz"))
    model, hw, bs, seed, seg = CASES[name]
    x = torch.from_numpy(detgen.scene((bs, 3, hw, hw), seed=seed))
    return g, yo.model_cfg(model), x, seed, seg


def state_dict(name, g, fused):
    model, _, _, seed, _ = CASES[name]
    aff = [(g[f"head_scale{i}"], g[f"head_bias{i}"]) for i in range(3)]
    gs = float(g["bn_gamma_scale"]) if "bn_gamma_scale" in g.files else None
    return yo.det_state_dict(yo.model_cfg(model), seed, fused=fused, bn_stats=(g["bn_mean"], g["bn_var"]), head_affine=aff, bn_gamma_scale=gs)


def box_iou(a, b):
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); bb = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + bb[None, :] - inter + 1e-9)


def agreement(ref, got, conf_thres, box_atol=2.0, conf_atol=0.02, margin=0.01):
    """Detection-set agreement between two NMS outputs (k, 6+) [x1,y1,x2,y2,conf,cls]: every reference detection whose confidence
    clears the threshold by `margin` must have a partner of the same class with all four corners within box_atol px and confidence
    within conf_atol -- and vice versa.  Detections inside the margin around the confidence threshold may legitimately appear on
    one side only (their kept / dropped decision flips within fp16 resolution), as may detections whose NMS decision was a near tie
    (an IoU within fp16 noise of the NMS threshold against a kept box); those are counted and bounded by the caller.
    Returns dict(matched, ref_strong, got_strong, unmatched_ref, unmatched_got, max_box_err, max_conf_err)."""
    out = dict(matched=0, unmatched_ref=0, unmatched_got=0, max_box_err=0.0, max_conf_err=0.0)
    rs = ref[ref[:, 4] >= conf_thres + margin]
    gs = got[got[:, 4] >= conf_thres + margin]
    out["ref_strong"], out["got_strong"] = len(rs), len(gs)
    for a, b, key in ((rs, got, "unmatched_ref"), (gs, ref, "unmatched_got")):
        if len(a) == 0:
            continue
        if len(b) == 0:
            out[key] += len(a)
            continue
        dist = np.abs(a[:, None, :4] - b[None, :, :4]).max(2)                     # (na, nb) worst corner distance
        dist = np.where(a[:, None, 5] == b[None, :, 5], dist, np.inf)
        dist = np.where(np.abs(a[:, None, 4] - b[None, :, 4]) <= conf_atol, dist, np.inf)
        j = dist.argmin(1)
        be = dist[np.arange(len(a)), j]
        good = be <= box_atol
        out[key] += int((~good).sum())
        if key == "unmatched_ref":
            out["matched"] = int(good.sum())
            if good.any():
                out["max_box_err"] = float(be[good].max())
                out["max_conf_err"] = float(np.abs(a[good, 4] - b[j[good], 4]).max())
    return out
