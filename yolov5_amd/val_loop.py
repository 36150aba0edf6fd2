"""The reference's validation loop, composed from the yolov5_amd seams (val.py:202-218 set-up, :255-333 batch loop and metrics, :390-396 return
value) -- what train.py calls once per epoch on the EMA model (train.py:440-455) and what `val.py` runs stand-alone.  Not the CLI: no dataset
yaml, plots, txt / json export, confusion matrix or callbacks; the numbers the caller gets back are the reference's

    (mp, mr, map50, map, box_loss, obj_loss, cls_loss), maps, (pre-process, inference, NMS ms per image)

MI355X mapping, per batch: uint8 batch -> the engine's input kernel (/255 fused) -> forward plan -> `non_max_suppression(..., multi_label=True,
padded=True)` (no host sync) -> `ValStats.update` = ONE matching launch for the batch (`y5_val_match`: both `scale_boxes` calls of val.py:298,304,
`xywh2xyxy`, `process_batch` for the ten IoU thresholds) -- the per-image Python loop of val.py:282-309 does not exist, and the statistics stay
on the device until `compute()` (one D2H copy per validation run).  `ap_per_class` runs on the host as in the reference (utils/metrics.py:25-126).
"""
from __future__ import annotations

import time

import numpy as np
import torch

from .general import non_max_suppression
from .metrics import ValStats


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def run(model, dataloader, conf_thres=0.001, iou_thres=0.6, max_det=300, half=True, single_cls=False, compute_loss=None, nc=None,
        training=True, profile=False):
    """val.py:run for a model that is already on the device.

    dataloader yields (im uint8|float BCHW, targets (M, 6) [img, cls, x, y, w, h] normalised, paths, shapes) with shapes[i] =
    ((h0, w0), ((gain_h, gain_w), (pad_w, pad_h))) as the reference's `LoadImagesAndLabels` returns them, or None (boxes compared in the
    letterboxed frame).  `half`: the model is converted IN PLACE for the run and back to float afterwards, exactly as val.py:187,388 do to the
    EMA model train.py passes in (its weights therefore pick up one fp16 rounding per validation -- reference behaviour, kept).
    `profile=True` synchronises around the three phases to fill the per-image times (val.py's `Profile`); off, the loop never blocks the host
    between batches and the times are the wall clock of the whole run split by enqueue time."""
    device = next(model.parameters()).device
    was_training = model.training
    model.half() if half else model.float()                                   # val.py:187
    model.eval()                                                              # :212
    det = model.model[-1] if hasattr(model, "model") else None
    if nc is None:
        nc = 1 if single_cls else int(getattr(det, "nc", 80))                 # :215
    iouv = torch.linspace(0.5, 0.95, 10, device=device)                       # :216
    stats = ValStats(iouv)
    loss = torch.zeros(3, device=device)                                      # :232
    dt = [0.0, 0.0, 0.0]
    nb_batches = 0
    for im, targets, paths, shapes in dataloader:                             # :237
        t0 = time.perf_counter()
        im = im.to(device, non_blocking=True)                                 # :240-243
        targets = targets.to(device).clone()
        if im.dtype != torch.uint8:                                           # uint8 goes in as it is: the input kernel divides by 255 (:244-245)
            im = im.half() if half else im.float()
        nb, _, height, width = im.shape
        if profile:
            _sync(device)
        t1 = time.perf_counter()
        out = model(im)                                                       # :250
        preds, train_out = out[0], (out[-1] if len(out) > 1 else None)
        if compute_loss is not None and train_out is not None:
            loss += compute_loss(train_out, targets)[1]                       # :253-254 (box, obj, cls)
        if profile:
            _sync(device)
        t2 = time.perf_counter()
        targets[:, 2:] *= torch.tensor((width, height, width, height), device=device, dtype=targets.dtype)   # :257 to pixels
        det_out, cnt = non_max_suppression(preds, conf_thres, iou_thres, multi_label=True, agnostic=single_cls, max_det=max_det, padded=True)  # :260-263
        if single_cls:
            det_out[..., 5] = 0                                               # :282
        stats.update(det_out, cnt, targets, shapes)                           # :266-309 for the whole batch in one launch
        if profile:
            _sync(device)
        t3 = time.perf_counter()
        dt[0] += t1 - t0
        dt[1] += t2 - t1
        dt[2] += t3 - t2
        nb_batches += 1
    res = stats.compute(nc)                                                   # :325-331
    mp, mr, map50, map_ = res["mp"], res["mr"], res["map50"], res["map"]
    maps = np.zeros(nc) + map_                                                # :392-394
    ap = res["ap"].mean(1) if len(res["ap"]) else np.zeros(0)
    for i, c in enumerate(res["ap_class"]):
        maps[int(c)] = ap[i]
    seen = max(stats.seen, 1)
    t = tuple(x / seen * 1e3 for x in dt)                                     # :346 ms per image
    model.float()                                                             # :388 "for training"
    if was_training:
        model.train()
    losses = (loss.cpu() / max(nb_batches, 1)).tolist()                       # :395
    return (mp, mr, map50, map_, *losses), maps, t
