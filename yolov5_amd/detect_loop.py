"""The reference's detection loop, composed from the yolov5_amd seams (detect.py:201-260): per source image
`letterbox -> CHW / RGB / float / 255 -> model -> non_max_suppression -> scale_boxes(...).round()`, here batched on the device:

  * images are decoded on the host (PIL; the reference uses cv2.imread, BGR -- `bgr=True` for arrays that come from cv2);
  * letterbox + layout + normalisation of a whole batch is ONE launch over the raw uint8 frames (augmentations.letterbox_batch:
    the host computes only utils/augmentations.py:85-115's geometry), the model forward is one plan replay, NMS is one kernel chain
    for all images and the de-letterboxing of all detections (general.py:613-626, detect.py:248 with .round()) one more launch;
  * `DetectPipeline` keeps the GPU fed across batches: the forward of batch i+1 is queued behind the NMS of batch i before the host
    waits for batch i's per-image counts (one stream, one-deep deferral of the only host sync).

Not the CLI: no argparse, video / stream sources, annotator or file writers."""
from __future__ import annotations

import os

import numpy as np
import torch

from .augmentations import letterbox_batch
from .general import non_max_suppression, scale_boxes_batch


def load_image(path):
    """HWC uint8 RGB array of an image file (detect.py's LoadImages decodes with cv2.imread -> BGR; the channel order is a flag here)."""
    from PIL import Image

    with Image.open(path) as im:
        return np.array(im.convert("RGB"))  # (a writable copy: torch.from_numpy wants one)


def _to_device_frames(images, device):
    return [torch.from_numpy(np.ascontiguousarray(im)).to(device) if not torch.is_tensor(im) else im.to(device) for im in images]


@torch.no_grad()
def detect(model, images, imgsz=640, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic_nms=False, max_det=1000, bgr=False,
           auto=False, stride=32, half=None, batch_size=None):
    """images: list of HWC uint8 arrays / tensors (or file paths).  Returns a list (one entry per image) of (k, 6) fp32 CPU tensors
    [x1, y1, x2, y2, conf, cls] in ORIGINAL image pixels, rounded like detect.py:248.  model: DetectionModel / DetectMultiBackend."""
    inner = getattr(model, "model", model) if hasattr(model, "pt") else model
    p = next(inner.parameters())
    device = p.device
    dtype = torch.float16 if (half if half is not None else p.dtype == torch.float16) else torch.float32
    imgs = [load_image(im) if isinstance(im, (str, os.PathLike)) else im for im in images]
    if isinstance(imgsz, int):
        imgsz = (imgsz, imgsz)
    out = []
    bs = batch_size or len(imgs)
    for b0 in range(0, len(imgs), bs):
        frames = _to_device_frames(imgs[b0:b0 + bs], device)
        # rect inference (`auto=True`, dataloaders.py LoadImages) gives per-image shapes; one batch needs one shape
        x, shapes = letterbox_batch(frames, imgsz, auto=auto and len(frames) == 1, stride=stride, dtype=dtype, swap_rb=bgr)
        y = model(x)
        pred = y[0] if isinstance(y, (list, tuple)) else y
        det, cnt = non_max_suppression(pred, conf_thres, iou_thres, classes, agnostic_nms, max_det=max_det, padded=True)
        # detect.py:248 calls scale_boxes WITHOUT ratio_pad: gain / pad are recomputed from the two shapes (general.py:615-617), which
        # differs from the letterbox's own (rounded) padding by a fraction of a pixel -- val.py:298 is the caller that passes it
        scale_boxes_batch(tuple(x.shape[2:]), det, cnt, [s[0] for s in shapes], None, round_=True)
        counts = cnt.tolist()
        out += [det[i, :counts[i]].cpu() for i in range(len(frames))]
    return out


class DetectPipeline:
    """One-deep software pipeline over batches that are already resident in HBM: `submit(x)` enqueues the forward of `x`, its NMS and the copy
    of the per-image counts into pinned host memory, THEN waits for the previous batch's counts and returns the previous batch's result (None
    for the first call); `flush()` returns the last one.  Every batch goes through exactly the kernels of `non_max_suppression(model(x)[0])`;
    what changes is where the host waits and where the NMS runs:
      * the host: in the plain loop the GPU idles from the count copy of batch i until the host has woken up, built the result list and launched
        the next forward (~140 us of every 2.7 ms step in the rocprofv3 trace of bench.py) -- here the next forward is already queued;
      * `overlap=True` (default): the NMS chain of batch i runs on a HIGH-PRIORITY side stream, ordered behind forward i by an event, while the
        caller's stream goes straight on to forward i+1.  The greedy / sort kernels are one workgroup per image (64 of 256 CUs for ~80 us);
        with priority their workgroups take the first CU slots the forward's persistent kernels give back, and the rest of the chip keeps
        working: 2.58 -> 2.51 ms per step.  (The same side stream at NORMAL priority showed no gain: the starved NMS finished so late that the
        host wake-up gap came back -- profiles/r02/r02_pipeline_ab.log.)  Forward i+1 writes a different z block and the engine's other
        objectness plane, and forward i+2 is only enqueued after batch i has been collected, so nothing the NMS reads is overwritten under it."""

    def __init__(self, model, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, max_det=1000, nm=0, overlap=None):
        import os
        if overlap is None:
            from . import _lib
            overlap = not _lib.disabled("pipe_overlap")   # A/B switch (Y5_DISABLE=pipe_overlap): the NMS chain stays on the caller's stream
        self.model = model
        self.kw = dict(conf_thres=conf_thres, iou_thres=iou_thres, classes=classes, agnostic=agnostic, max_det=max_det, nm=nm)
        self.overlap = overlap
        self._side = None
        self._inflight = None  # (det, pinned counts, event: counts have landed, z kept alive until then, prototypes)
        self.protos = None     # SegmentationModel: the mask prototypes (bs, 32, H/4, W/4) of the batch the last submit() / flush() RETURNED
        self._pinned = []      # two host buffers, used alternately (one is being read while the other is being written)
        self._k = 0

    def _host_counts(self, n):
        if len(self._pinned) < 2 or self._pinned[0].numel() != n:
            self._pinned = [torch.empty((n,), dtype=torch.int32, pin_memory=True) for _ in range(2)]
        self._k ^= 1
        return self._pinned[self._k]

    def _post(self, z):
        det, cnt = non_max_suppression(z, padded=True, **self.kw)
        host = self._host_counts(cnt.numel())
        host.copy_(cnt, non_blocking=True)
        return det, host

    def submit(self, x):
        out = self.model(x)
        z = out[0]
        proto = out[1] if len(out) > 1 and torch.is_tensor(out[1]) else None  # segment/predict.py:139: pred, proto = model(im)[:2]
        cur = torch.cuda.current_stream(z.device)
        if self.overlap:
            if self._side is None:
                self._side = torch.cuda.Stream(z.device, priority=-1)
            fwd_done = torch.cuda.Event()
            fwd_done.record(cur)
            self._side.wait_event(fwd_done)
            with torch.cuda.stream(self._side):
                det, host = self._post(z)
                ev = torch.cuda.Event()
                ev.record(self._side)
            # allocator bookkeeping across the two streams: the results were allocated on the side stream and are consumed by the caller on
            # `cur` (scale_boxes, matching); z was allocated on `cur` and is read on the side stream.  Without this a freed block could be
            # handed to the next side-stream NMS while consumer kernels on `cur` are still queued.
            det.record_stream(cur)
            z.record_stream(self._side)
        else:
            det, host = self._post(z)
            ev = torch.cuda.Event()
            ev.record(cur)
        prev, self._inflight = self._inflight, (det, host, ev, z, proto)
        return self._collect(prev)

    def flush(self):
        prev, self._inflight = self._inflight, None
        return self._collect(prev)

    def _collect(self, prev):
        if prev is None:
            return None
        det, host, ev, _z, self.protos = prev
        ev.synchronize()                        # the one host wait per batch; the GPU already has the next batch queued
        counts = host.tolist()
        bs, max_det, w = det.shape
        sizes = [v for c in counts for v in (c, max_det - c)]   # one split call instead of bs slicings (general.non_max_suppression)
        return list(det.view(bs * max_det, w).split(sizes)[::2])
