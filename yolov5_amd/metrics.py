"""Validation metrics with the reference's names.

Device side (one HIP launch per batch, `csrc/metrics.hip` -> `y5_val_match`):
    process_batch(detections, labels, iouv)            utils/metrics.py:224-265 (box branch)
    match_batch(out, counts, targets, shapes, iouv)    val.py:282-307 for all images of a batch: de-letterbox of the
                                                        predictions and labels fused with the matching
    ValStats                                            the `stats` list of val.py:218,308 kept on the device
Host side -- numpy, as in the reference (SURVEY keeps `ap_per_class` on the CPU: it runs once per epoch on a few
thousand rows): `ap_per_class`, `compute_ap`, `smooth`, `fitness`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _need_gpu(t, what):
    if not _lib.accepts(t):
        raise RuntimeError(f"yolov5_amd.metrics.{what} needs GPU tensors (no CPU path)")


def process_batch(detections, labels, iouv):
    """utils/metrics.py:224-265: detections (N,6+) [x1,y1,x2,y2,conf,cls], labels (M,5) [cls,x1,y1,x2,y2], iouv (niou)
    -> bool (N, niou) on the device.  Tie rule and the equivalence with the reference's sort are in csrc/metrics.hip."""
    _need_gpu(detections, "process_batch")
    n = detections.shape[0]
    niou = iouv.numel()
    if n == 0:
        return torch.zeros((0, niou), dtype=torch.bool, device=detections.device)
    if n > 1024:
        raise ValueError("process_batch: at most 1024 detections per image")
    det = detections.float().contiguous()
    lab = labels.float().contiguous()
    iv = iouv.to(device=det.device, dtype=torch.float32).contiguous()
    correct = torch.empty((1, n, niou), dtype=torch.uint8, device=det.device)
    lib = _lib.lib()
    rc = lib.y5_val_match(_p(det), det.shape[1], n, None, 1, _p(lab) if lab.numel() else None, 5, lab.shape[0], -1, 0, 1, 0, None, _p(iv), niou,
                          _p(correct), None, _lib.stream(det.device))
    _lib.check(rc, lib)
    return correct[0].bool()


def match_batch(out, counts, targets, shapes, iouv, predn=False):
    """val.py:282-307 for the whole batch in one launch.

    out (bs, max_det, 6+nm) / counts (bs) int32: the padded NMS result (`general.non_max_suppression(..., padded=True)`);
    targets (M,6) [img, cls, cx, cy, w, h] in letterboxed pixels (after val.py:274); shapes: the dataloader's per-image
    ((h0, w0), ((gain, gain), (pad_x, pad_y))) or None to compare the boxes as given.
    Returns correct (bs, max_det, niou) uint8 [rows past counts are 0] and, with predn=True, the native-space boxes
    (bs, max_det, 4) of val.py:297-298."""
    _need_gpu(out, "match_batch")
    bs, max_det, ld = out.shape
    dev = out.device
    if out.dtype != torch.float32 or not out.is_contiguous():
        out = out.float().contiguous()
    tg = targets.to(device=dev, dtype=torch.float32).contiguous()
    iv = iouv.to(device=dev, dtype=torch.float32).contiguous()
    niou = iv.numel()
    sc = None
    if shapes is not None:
        sc = torch.tensor([[rp[0][0], rp[1][0], rp[1][1], s0[0], s0[1]] for s0, rp in shapes], dtype=torch.float32).to(dev, non_blocking=True)
    correct = torch.empty((bs, max_det, niou), dtype=torch.uint8, device=dev)
    pn = torch.empty((bs, max_det, 4), dtype=torch.float32, device=dev) if predn else None
    cn = counts.to(device=dev, dtype=torch.int32) if counts is not None else None
    lib = _lib.lib()
    rc = lib.y5_val_match(_p(out), ld, max_det, _p(cn), bs, _p(tg) if tg.numel() else None, 6, tg.shape[0], 0, 1, 2, 1, _p(sc), _p(iv), niou,
                          _p(correct), _p(pn), _lib.stream(dev))
    _lib.check(rc, lib)
    return (correct, pn) if predn else correct


class ValStats:
    """The `stats` accumulator of val.py:218,308-309,325: (correct, conf, pcls, tcls) per batch, kept on the device until
    `compute()`; one D2H copy per validation run instead of one per image."""

    def __init__(self, iouv):
        self.iouv = iouv
        self.correct, self.conf, self.pcls, self.tcls = [], [], [], []
        self.seen = 0

    def update(self, out, counts, targets, shapes=None):
        """out/counts: padded NMS result of one batch; targets (M,6) in letterboxed pixels; shapes as in match_batch."""
        bs, max_det, _ = out.shape
        correct = match_batch(out, counts, targets, shapes, self.iouv)
        valid = torch.arange(max_det, device=out.device)[None, :] < counts.to(out.device)[:, None]
        self.correct.append(correct[valid].bool())
        self.conf.append(out[..., 4][valid])
        self.pcls.append(out[..., 5][valid])
        self.tcls.append(targets[:, 1].to(out.device))
        self.seen += bs

    def compute(self, nc=None):
        """val.py:325-330 -> dict(mp, mr, map50, map, ap, ap_class, nt)."""
        if not self.correct:
            return dict(mp=0.0, mr=0.0, map50=0.0, map=0.0, ap=np.zeros((0, self.iouv.numel())), ap_class=np.zeros(0, int), nt=np.zeros(0, int))
        tp = torch.cat(self.correct).cpu().numpy()
        conf = torch.cat(self.conf).cpu().numpy()
        pcls = torch.cat(self.pcls).cpu().numpy()
        tcls = torch.cat(self.tcls).cpu().numpy()
        res = dict(mp=0.0, mr=0.0, map50=0.0, map=0.0, ap=np.zeros((0, tp.shape[1])), ap_class=np.zeros(0, int))
        if tp.shape[0] and tp.any():
            _, _, p, r, _, ap, ap_class = ap_per_class(tp, conf, pcls, tcls)
            res.update(mp=float(p.mean()), mr=float(r.mean()), map50=float(ap[:, 0].mean()), map=float(ap.mean()), ap=ap, ap_class=ap_class)
        res["nt"] = np.bincount(tcls.astype(int), minlength=nc or 0)
        return res


# ----------------------------------------------------------------------------------------------------------------------
# host side (numpy; CPU in the reference as well)
# ----------------------------------------------------------------------------------------------------------------------
def fitness(x):
    """utils/metrics.py:19-22: 0.1 * mAP@0.5 + 0.9 * mAP@0.5:0.95 of rows [P, R, mAP@0.5, mAP@0.5:0.95, ...]."""
    return (np.asarray(x)[:, :4] * np.array([0.0, 0.0, 0.1, 0.9])).sum(1)


def smooth(y, f=0.05):
    """ultralytics.utils.metrics.smooth (used at utils/metrics.py:91): moving average over an odd window of about
    2*f*len(y) samples, the ends extended with the first / last value."""
    taps = round(len(y) * f * 2) // 2 + 1
    half = taps // 2
    padded = np.concatenate((np.full(half, y[0], dtype=float), y, np.full(half, y[-1], dtype=float)))
    return np.convolve(padded, np.full(taps, 1.0 / taps), mode="valid")


def compute_ap(recall, precision):
    """utils/metrics.py:98-126 ('interp'): area under the monotone precision envelope sampled at 101 recall points."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    ap = ((y[1:] + y[:-1]) * (x[1:] - x[:-1]) / 2.0).sum()
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, save_dir=".", names=(), eps=1e-16, prefix=""):
    """utils/metrics.py:25-95.  tp (n, niou) bool, conf (n), pred_cls (n), target_cls (m) ->
    tp, fp, p, r, f1 (per class at the best mean-F1 confidence), ap (nc, niou), unique classes.  Plots are not produced."""
    if plot:
        raise NotImplementedError("ap_per_class(plot=True): PR / F1 curve plots are outside the hot path")
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, n_labels = np.unique(target_cls, return_counts=True)
    nc, niou = classes.shape[0], tp.shape[1]
    grid = np.linspace(0, 1, 1000)
    ap = np.zeros((nc, niou))
    p_curve = np.zeros((nc, 1000))
    r_curve = np.zeros((nc, 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        if not sel.any() or n_labels[ci] == 0:
            continue
        hits = tp[sel]
        tpc = hits.cumsum(0)
        fpc = (1 - hits).cumsum(0)
        recall = tpc / (n_labels[ci] + eps)
        precision = tpc / (tpc + fpc)
        neg_conf = -conf[sel]
        r_curve[ci] = np.interp(-grid, neg_conf, recall[:, 0], left=0)
        p_curve[ci] = np.interp(-grid, neg_conf, precision[:, 0], left=1)
        for j in range(niou):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    best = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, best], r_curve[:, best], f1_curve[:, best]
    tp_n = (r * n_labels).round()
    fp_n = (tp_n / (p + eps) - tp_n).round()
    return tp_n, fp_n, p, r, f1, ap, classes.astype(int)
