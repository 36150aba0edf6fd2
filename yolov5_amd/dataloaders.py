"""Training input pipeline (utils/dataloaders.py:696-863 `LoadImagesAndLabels.__getitem__` with augment = True, rect = False -- the mosaic
branch and, where the hyp['mosaic'] gate sends a sample there, the letterbox branch :710-733 -- + `collate_fn`;
utils/augmentations.py:69-83 `augment_hsv`, :118-209 `random_perspective`, :246-258 `box_candidates`) with the pixel work on the
device: the dataset's uint8 BGR images live in HBM, `MosaicLoader` draws the random numbers and does the geometry + label transform
of a whole batch on the host (a few hundred floats) and ONE `y5_mosaic_batch` launch renders the (B, 3, s, s) batch -- resized
tiles, canvas, warp, HSV, flips, layout -- directly in the a0 tensor contract of the model (uint8, or fp16 / 255).  At the training
rates of this engine (3 k img/s per GPU) the reference's 8 cv2 worker processes cannot keep a GPU fed.

Random draws follow the reference's ORDER per sample (documented in `draw_sample`), so that a `random.seed()` / `np.random.seed()`-ed
run consumes the generators exactly as `__getitem__` does; they can also be passed in (parity tests).
Not covered: rect batches / augment = False (the validation loader: augmentations.letterbox_batch), copy_paste, albumentations,
perspective != 0, segments."""
from __future__ import annotations

import ctypes as C
import math
import random

import numpy as np
import torch

from . import _lib

HYP_AUG = {"hsv_h": 0.015, "hsv_s": 0.7, "hsv_v": 0.4, "degrees": 0.0, "translate": 0.1, "scale": 0.5, "shear": 0.0, "perspective": 0.0,
           "flipud": 0.0, "fliplr": 0.5, "mosaic": 1.0, "mixup": 0.0, "copy_paste": 0.0}  # data/hyps/hyp.scratch-low.yaml:25-37


def draw_sample(index, n_images, s, hyp, rng=random, np_rng=np.random):
    """Random numbers of one sample in the reference's order: mosaic gate (dataloaders.py:701); centre yc, xc (:802); three extra
    indices + shuffle (:803-804); perspective x2, angle, scale, shear x2, translate x2 (augmentations.py:135-156); mixup gate
    (dataloaders.py:707); three HSV gains from numpy (augmentations.py:72); flipud, fliplr gates (dataloaders.py:747,753)."""
    d = {"mosaic": rng.random() < hyp["mosaic"]}
    if d["mosaic"]:
        d["yc"], d["xc"] = (int(rng.uniform(-x, 2 * s + x)) for x in (-s // 2, -s // 2))
        idx = [index, *rng.choices(range(n_images), k=3)]
        rng.shuffle(idx)
        d["indices"] = idx
    else:   # letterbox branch (dataloaders.py:710-733): the next draws are random_perspective's; no mixup gate on this side
        d["indices"] = [index]
    d["persp"] = (rng.uniform(-hyp["perspective"], hyp["perspective"]), rng.uniform(-hyp["perspective"], hyp["perspective"]))
    d["angle"] = rng.uniform(-hyp["degrees"], hyp["degrees"])
    d["scale"] = rng.uniform(1 - hyp["scale"], 1 + hyp["scale"])
    d["shear"] = (rng.uniform(-hyp["shear"], hyp["shear"]), rng.uniform(-hyp["shear"], hyp["shear"]))
    d["translate"] = (rng.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]), rng.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]))
    if d["mosaic"] and rng.random() < hyp["mixup"]:
        # dataloaders.py:707-708: random.choice(indices), the partner mosaic's own draws (load_mosaic -> random_perspective), np.random.beta in mixup()
        m = {"mosaic": True}
        i2 = rng.choice(range(n_images))
        m["yc"], m["xc"] = (int(rng.uniform(-x, 2 * s + x)) for x in (-s // 2, -s // 2))
        idx = [i2, *rng.choices(range(n_images), k=3)]
        rng.shuffle(idx)
        m["indices"] = idx
        m["persp"] = (rng.uniform(-hyp["perspective"], hyp["perspective"]), rng.uniform(-hyp["perspective"], hyp["perspective"]))
        m["angle"] = rng.uniform(-hyp["degrees"], hyp["degrees"])
        m["scale"] = rng.uniform(1 - hyp["scale"], 1 + hyp["scale"])
        m["shear"] = (rng.uniform(-hyp["shear"], hyp["shear"]), rng.uniform(-hyp["shear"], hyp["shear"]))
        m["translate"] = (rng.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]), rng.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]))
        d["partner"], d["mix_r"] = m, float(np_rng.beta(32.0, 32.0))
    d["hsv"] = np_rng.uniform(-1, 1, 3) * [hyp["hsv_h"], hyp["hsv_s"], hyp["hsv_v"]] + 1
    d["flipud"] = rng.random() < hyp["flipud"]
    d["fliplr"] = rng.random() < hyp["fliplr"]
    if hyp["perspective"]:
        raise NotImplementedError("MosaicLoader: perspective != 0 (cv2.warpPerspective) is not implemented")
    return d


def _resized_hw(h0, w0, s):
    """dataloaders.py:783-787."""
    r = s / max(h0, w0)
    return (h0, w0) if r == 1 else (math.ceil(h0 * r), math.ceil(w0 * r))


def _tile_rects(hw, yc, xc, s):
    """Canvas rectangle [x1a, x2a) x [y1a, y2a) and source offset (x1b, y1b) of the 4 tiles around (xc, yc) (dataloaders.py:810-822)."""
    rects = []
    for t, (h, w) in enumerate(hw):
        left, top = t in (0, 2), t in (0, 1)
        x1a, x2a = (max(xc - w, 0), xc) if left else (xc, min(xc + w, 2 * s))
        y1a, y2a = (max(yc - h, 0), yc) if top else (yc, min(2 * s, yc + h))
        x1b = w - (x2a - x1a) if left else 0
        y1b = h - (y2a - y1a) if top else 0
        rects.append((x1a, y1a, x2a, y2a, x1b, y1b))
    return rects


def _affine(d, s, mosaic=True):
    """M = T S R P C of random_perspective (augmentations.py:124-160) -> (M 3x3, out w, h): the 2s x 2s mosaic canvas with border (-s/2, -s/2), or
    the s x s letterboxed image of the non-mosaic branch with border (0, 0)."""
    src = 2 * s if mosaic else s
    height = width = src + (2 * (-s // 2) if mosaic else 0)
    Cm = np.eye(3)
    Cm[0, 2] = Cm[1, 2] = -src / 2
    a = d["angle"] * math.pi / 180.0
    ca, sa = math.cos(a) * d["scale"], math.sin(a) * d["scale"]
    R = np.array([[ca, sa, 0.0], [-sa, ca, 0.0], [0.0, 0.0, 1.0]])      # cv2.getRotationMatrix2D(angle, (0, 0), scale)
    Sh = np.eye(3)
    Sh[0, 1] = math.tan(d["shear"][0] * math.pi / 180)
    Sh[1, 0] = math.tan(d["shear"][1] * math.pi / 180)
    T = np.eye(3)
    T[0, 2], T[1, 2] = d["translate"][0] * width, d["translate"][1] * height
    return T @ Sh @ R @ np.eye(3) @ Cm, width, height


def _invert_affine(M):
    """The inverse map cv::warpAffine builds from the forward matrix (same operation order, double precision)."""
    m = np.array(M[:2], dtype=np.float64).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0] = a11
    m[0, 1] *= -D
    m[1, 0] *= -D
    m[1, 1] = a22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    return m


def _labels(labels, d, hw, rects, M, width, height, s, pads=None):
    """Label half of the sample up to random_perspective: tiles -> canvas pixels (xywhn2xyxy, :838 / :720), mosaic only: clip to the canvas
    (:845-846), corners through M + hull + clip + box_candidates (augmentations.py:193-209,246-258) -> (k, 5) [cls, x1, y1, x2, y2] in output pixels.
    pads: the (padw, padh) of each tile when they are not the integer tile offsets (letterbox branch: the FLOAT half-borders dw, dh)."""
    parts = []
    for k, (i, (h, w), (x1a, y1a, _x2a, _y2a, x1b, y1b)) in enumerate(zip(d["indices"], hw, rects)):
        lb = np.array(labels[i], dtype=np.float32).reshape(-1, 5).copy()
        if lb.size:
            xy, half = lb[:, 1:3].copy(), lb[:, 3:5] / 2
            padw, padh = (x1a - x1b, y1a - y1b) if pads is None else pads[k]
            lb[:, 1] = w * (xy[:, 0] - half[:, 0]) + padw
            lb[:, 2] = h * (xy[:, 1] - half[:, 1]) + padh
            lb[:, 3] = w * (xy[:, 0] + half[:, 0]) + padw
            lb[:, 4] = h * (xy[:, 1] + half[:, 1]) + padh
        parts.append(lb)
    t = np.concatenate(parts, 0)
    if d.get("mosaic", True):
        np.clip(t[:, 1:], 0, 2 * s, out=t[:, 1:])
    n = len(t)
    if n:
        pts = np.ones((n * 4, 3))
        pts[:, :2] = t[:, [1, 2, 3, 4, 1, 4, 3, 2]].reshape(n * 4, 2)
        pts = (pts @ M.T)[:, :2].reshape(n, 8)
        xs, ys = pts[:, [0, 2, 4, 6]], pts[:, [1, 3, 5, 7]]
        new = np.concatenate((xs.min(1), ys.min(1), xs.max(1), ys.max(1))).reshape(4, n).T
        new[:, [0, 2]] = new[:, [0, 2]].clip(0, width)
        new[:, [1, 3]] = new[:, [1, 3]].clip(0, height)
        b1 = t[:, 1:5].T * d["scale"]
        w1, h1 = b1[2] - b1[0], b1[3] - b1[1]
        w2, h2 = new[:, 2] - new[:, 0], new[:, 3] - new[:, 1]
        ar = np.maximum(w2 / (h2 + 1e-16), h2 / (w2 + 1e-16))
        keep = (w2 > 2) & (h2 > 2) & (w2 * h2 / (w1 * h1 + 1e-16) > 0.10) & (ar < 100)
        t = t[keep]
        t[:, 1:5] = new[keep]
    return t


def _finish_labels(t, d, width, height):
    """dataloaders.py:735-757 on the (possibly mixup-concatenated) pixel boxes: normalised xywh with clipping (:737), flips."""
    if len(t):
        b = t[:, 1:5]
        b[:, [0, 2]] = b[:, [0, 2]].clip(0, width - 1e-3)
        b[:, [1, 3]] = b[:, [1, 3]].clip(0, height - 1e-3)
        out = b.copy()
        out[:, 0] = ((b[:, 0] + b[:, 2]) / 2) / width
        out[:, 1] = ((b[:, 1] + b[:, 3]) / 2) / height
        out[:, 2] = (b[:, 2] - b[:, 0]) / width
        out[:, 3] = (b[:, 3] - b[:, 1]) / height
        t[:, 1:5] = out
        if d["flipud"]:
            t[:, 2] = 1 - t[:, 2]
        if d["fliplr"]:
            t[:, 1] = 1 - t[:, 1]
    res = np.zeros((len(t), 6), dtype=np.float32)
    if len(t):
        res[:, 1:] = t
    return res


def mosaic_batch(images, labels, draws, s, hyp=None, dtype=torch.uint8, normalize=False):
    """Render one training batch.  images: list of uint8 (h, w, 3) BGR tensors resident on the device (any sizes); labels: list of
    (k, 5) arrays [cls, xc, yc, w, h] normalised; draws: list (one per output image) of `draw_sample` dicts.
    Returns (imgs (B, 3, s, s) `dtype` RGB CHW, targets (nt, 6) float32 [image index in batch, cls, xc, yc, w, h])."""
    hyp = HYP_AUG if hyp is None else hyp
    dev = images[0].device
    B = len(draws)
    labs = []
    use_hsv = bool(hyp["hsv_h"] or hyp["hsv_s"] or hyp["hsv_v"])
    x = np.arange(0, 256, dtype=np.float64)
    def fill(j, d):
        """Geometry half of one job (tiles, rectangles, inverse affine map) -> the pixel boxes of its labels after random_perspective."""
        hw = []
        for t, i in enumerate(d["indices"]):
            im = images[i]
            if not (_lib.accepts(im) and im.dtype == torch.uint8 and im.ndim == 3 and im.shape[2] == 3 and im.stride(2) == 1 and im.stride(1) == 3):
                raise ValueError("mosaic_batch: images must be uint8 (h, w, 3) device tensors with contiguous rows")
            h0, w0 = int(im.shape[0]), int(im.shape[1])
            rh, rw = _resized_hw(h0, w0, s)
            hw.append((rh, rw))
            j.src[t], j.h0[t], j.w0[t], j.stride[t], j.rh[t], j.rw[t] = im.data_ptr(), h0, w0, int(im.stride(0)), rh, rw
        mosaic, pads = d.get("mosaic", True), None
        if mosaic:
            rects = _tile_rects(hw, d["yc"], d["xc"], s)
        else:
            # letterbox(auto=False, scaleup=True) of an image whose longest side already is s (augmentations.py:85-115): r = 1, no second resize;
            # the image sits at (left, top) = (round(dw - 0.1), round(dh - 0.1)) of an s x s canvas of 114s, the labels move by the FLOAT dw, dh
            (rh, rw), = hw
            dw, dh = (s - rw) / 2, (s - rh) / 2
            left, top = int(round(dw - 0.1)), int(round(dh - 0.1))
            rects, pads = [(left, top, left + rw, top + rh, 0, 0)], [(dw, dh)]
            j.canvas = s
        for t, (x1a, y1a, x2a, y2a, x1b, y1b) in enumerate(rects):
            j.x1a[t], j.y1a[t], j.x2a[t], j.y2a[t], j.x1b[t], j.y1b[t] = x1a, y1a, x2a, y2a, x1b, y1b
        M, width, height = _affine(d, s, mosaic)
        A = _invert_affine(M)
        for k in range(6):
            j.A[k] = float(A.reshape(-1)[k])
        return _labels(labels, d, hw, rects, M, width, height, s, pads), width, height

    partners = [d for d in draws if d.get("partner") is not None]
    jobs = (_lib.MosaicJob * (B + len(partners)))()
    npart = 0
    for b, d in enumerate(draws):
        j = jobs[b]
        t, width, height = fill(j, d)
        if d.get("partner") is not None:   # mixup (dataloaders.py:707-708): the partner mosaic is a job of its own behind the B rendered ones
            t2, _, _ = fill(jobs[B + npart], d["partner"])
            j.mix_job, j.mix_r = B + npart + 1, float(d["mix_r"])
            npart += 1
            t = np.concatenate((t, t2), 0)
        r = np.asarray(d["hsv"], dtype=np.float64)
        luts = (((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8), np.clip(x * r[2], 0, 255).astype(np.uint8))
        for c in range(3):
            C.memmove(j.lut[c], luts[c].ctypes.data, 256)
        j.hsv, j.flipud, j.fliplr = int(use_hsv), int(bool(d["flipud"])), int(bool(d["fliplr"]))
        lb = _finish_labels(t, d, width, height)
        lb[:, 0] = b                                             # collate_fn (dataloaders.py:860-862)
        labs.append(lb)
    table = torch.frombuffer(bytearray(jobs), dtype=torch.uint8).to(dev)
    out = torch.empty((B, 3, s, s), dtype=dtype, device=dev)
    code = {torch.uint8: _lib.Y5_U8, torch.float16: _lib.Y5_F16, torch.float32: _lib.Y5_F32}[dtype]
    lib = _lib.lib()
    _lib.check(lib.y5_mosaic_batch(C.c_void_p(table.data_ptr()), B, s, 114, C.c_void_p(out.data_ptr()), code, int(normalize and dtype != torch.uint8),
                                   _lib.stream(dev)), lib)
    targets = torch.from_numpy(np.concatenate(labs, 0) if labs else np.zeros((0, 6), np.float32))
    return out, targets


class MosaicLoader:
    """Re-iterable training loader over an in-HBM dataset: every epoch visits the images in a fresh random order (DataLoader
    shuffle=True / SmartDistributedSampler: rank r takes indices r::world of the epoch's permutation), each batch is one
    `mosaic_batch` launch.  Yields (imgs, targets, paths, shapes) like the reference's collate_fn."""

    def __init__(self, images, labels, img_size=640, batch_size=16, hyp=None, dtype=torch.uint8, rank=-1, world_size=1, seed=0, paths=None):
        self.images, self.labels, self.s, self.bs = images, labels, img_size, batch_size
        self.hyp = dict(HYP_AUG if hyp is None else hyp)
        self.dtype, self.rank, self.world, self.seed, self.epoch = dtype, rank, world_size, seed, 0
        self.paths = paths or [f"image{i}" for i in range(len(images))]
        n = len(images)
        # SmartDistributedSampler pads every rank to num_samples = ceil(n / world) (utils/dataloaders.py:94-101): same batch count on all ranks
        self.n_local = (n + world_size - 1) // world_size if rank != -1 else n

    def set_epoch(self, epoch):
        self.epoch = epoch

    @property
    def sampler(self):
        return self

    def __len__(self):
        return (self.n_local + self.bs - 1) // self.bs

    def __iter__(self):
        g = random.Random(self.seed + self.epoch)
        order = list(range(len(self.images)))
        g.shuffle(order)
        if self.rank != -1:
            from .train_loop import pad_to_common

            order = pad_to_common(order[self.rank::self.world], len(self.images), self.world)
        for b0 in range(0, len(order), self.bs):
            ids = order[b0:b0 + self.bs]
            draws = [draw_sample(i, len(self.images), self.s, self.hyp) for i in ids]
            imgs, targets = mosaic_batch(self.images, self.labels, draws, self.s, self.hyp, self.dtype, normalize=True)
            yield imgs, targets, [self.paths[i] for i in ids], None
        self.epoch += 1
