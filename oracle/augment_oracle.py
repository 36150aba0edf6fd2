"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (yolov5_amd/).

CPU restatement (numpy, float64 geometry like the reference) of ONE training sample of the reference's input pipeline with every
random draw passed in: utils/dataloaders.py:696-760 `__getitem__` (mosaic branch), :770-790 `load_image`, :798-855 `load_mosaic`,
utils/augmentations.py:118-190 `random_perspective` (affine case), :246-258 `box_candidates`, :69-83 `augment_hsv`, flips and the
HWC-BGR -> CHW-RGB conversion, plus :858-863 `collate_fn`.  The OpenCV primitives underneath (resize, warpAffine, cvtColor, LUT)
are the restated ones of oracle/thirdparty.py (parity unpinned: opencv-python is absent); everything above them is pinned against
the reference's own functions through tests/golden/augment.npz (oracle/make_golden.py:gen_augment)."""
from __future__ import annotations

import math
import random

import numpy as np

from . import thirdparty as tp

HYP_AUG = {"hsv_h": 0.015, "hsv_s": 0.7, "hsv_v": 0.4, "degrees": 0.0, "translate": 0.1, "scale": 0.5, "shear": 0.0, "perspective": 0.0,
           "flipud": 0.0, "fliplr": 0.5, "mosaic": 1.0, "mixup": 0.0, "copy_paste": 0.0}  # data/hyps/hyp.scratch-low.yaml:25-37


def reference_draws(seed, index, n_images, s, hyp):
    """The random numbers one `__getitem__(index)` call of the reference consumes, in ITS order, from Python's `random` and numpy's
    global generator seeded with `seed` (mosaic branch, copy_paste = 0):
      random():            mosaic gate                                    dataloaders.py:701
      uniform x2:          mosaic centre yc, xc                           :802
      choices(k=3), shuffle                                               :803-804
      uniform x8:          perspective x2, angle, scale, shear x2, translate x2   augmentations.py:135-156
      random():            mixup gate                                     dataloaders.py:707
      np.random.uniform(-1, 1, 3): HSV gains                              augmentations.py:72
      random() x2:         flipud, fliplr gates                           dataloaders.py:747,753"""
    random.seed(seed)
    np.random.seed(seed)
    d = {"mosaic": random.random() < hyp["mosaic"]}
    if d["mosaic"]:
        border = [-s // 2, -s // 2]
        d["yc"], d["xc"] = (int(random.uniform(-x, 2 * s + x)) for x in border)
        idx = [index, *random.choices(range(n_images), k=3)]
        random.shuffle(idx)
        d["indices"] = idx
    else:  # letterbox branch (dataloaders.py:710-733): no draws before random_perspective's, and no mixup gate behind them
        d["indices"] = [index]
    d["persp"] = (random.uniform(-hyp["perspective"], hyp["perspective"]), random.uniform(-hyp["perspective"], hyp["perspective"]))
    d["angle"] = random.uniform(-hyp["degrees"], hyp["degrees"])
    d["scale"] = random.uniform(1 - hyp["scale"], 1 + hyp["scale"])
    d["shear"] = (random.uniform(-hyp["shear"], hyp["shear"]), random.uniform(-hyp["shear"], hyp["shear"]))
    d["translate"] = (random.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]), random.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]))
    d["mixup"] = d["mosaic"] and random.random() < hyp["mixup"]
    if d["mixup"]:  # dataloaders.py:707-708: random.choice(indices), the partner mosaic's own draws (load_mosaic), then np.random.beta inside mixup()
        i2 = random.choice(range(n_images))
        m = {"mosaic": True}
        border = [-s // 2, -s // 2]
        m["yc"], m["xc"] = (int(random.uniform(-x, 2 * s + x)) for x in border)
        idx = [i2, *random.choices(range(n_images), k=3)]
        random.shuffle(idx)
        m["indices"] = idx
        m["persp"] = (random.uniform(-hyp["perspective"], hyp["perspective"]), random.uniform(-hyp["perspective"], hyp["perspective"]))
        m["angle"] = random.uniform(-hyp["degrees"], hyp["degrees"])
        m["scale"] = random.uniform(1 - hyp["scale"], 1 + hyp["scale"])
        m["shear"] = (random.uniform(-hyp["shear"], hyp["shear"]), random.uniform(-hyp["shear"], hyp["shear"]))
        m["translate"] = (random.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]), random.uniform(0.5 - hyp["translate"], 0.5 + hyp["translate"]))
        d["partner"] = m
        d["mix_r"] = np.random.beta(32.0, 32.0)
    d["hsv"] = np.random.uniform(-1, 1, 3) * [hyp["hsv_h"], hyp["hsv_s"], hyp["hsv_v"]] + 1
    d["flipud"] = random.random() < hyp["flipud"]
    d["fliplr"] = random.random() < hyp["fliplr"]
    return d


def load_image(im, s, augment=True):
    """dataloaders.py:770-790: longest side -> s (INTER_LINEAR when augmenting or up-scaling)."""
    h0, w0 = im.shape[:2]
    r = s / max(h0, w0)
    if r != 1:
        assert augment or r > 1, "INTER_AREA (validation down-scale) is not restated"
        im = tp.cv2_resize(im, (math.ceil(w0 * r), math.ceil(h0 * r)), interpolation=1)
    return im, (h0, w0), im.shape[:2]


def mosaic_tiles(shapes_hw, yc, xc, s):
    """dataloaders.py:810-822: for the 4 tiles (resized h, w) -> [(x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b)] (canvas rect, source rect)."""
    out = []
    for i, (h, w) in enumerate(shapes_hw):
        if i == 0:
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
        elif i == 1:
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
            x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
        elif i == 2:
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
        else:
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
        out.append((x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b))
    return out


def perspective_matrix(d, src_hw, border):
    """augmentations.py:124-160: M = T @ S @ R @ P @ C for an input of src_hw and output (src + 2 * border)."""
    height, width = src_hw[0] + border[0] * 2, src_hw[1] + border[1] * 2
    C = np.eye(3)
    C[0, 2], C[1, 2] = -src_hw[1] / 2, -src_hw[0] / 2
    P = np.eye(3)
    P[2, 0], P[2, 1] = d["persp"]
    R = np.eye(3)
    R[:2] = tp.cv2_get_rotation_matrix_2d((0, 0), d["angle"], d["scale"])
    S = np.eye(3)
    S[0, 1] = math.tan(d["shear"][0] * math.pi / 180)
    S[1, 0] = math.tan(d["shear"][1] * math.pi / 180)
    T = np.eye(3)
    T[0, 2], T[1, 2] = d["translate"][0] * width, d["translate"][1] * height
    return T @ S @ R @ P @ C, width, height


def box_candidates(box1, box2, wh_thr=2, ar_thr=100, area_thr=0.1, eps=1e-16):
    """augmentations.py:246-258."""
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def warp_boxes(targets, M, width, height, scale):
    """augmentations.py:168-209 (box branch): 4 corners through M, axis-aligned hull, clip, candidate filter."""
    n = len(targets)
    if not n:
        return targets
    xy = np.ones((n * 4, 3))
    xy[:, :2] = targets[:, [1, 2, 3, 4, 1, 4, 3, 2]].reshape(n * 4, 2)
    xy = xy @ M.T
    xy = xy[:, :2].reshape(n, 8)
    x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
    new = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
    new[:, [0, 2]] = new[:, [0, 2]].clip(0, width)
    new[:, [1, 3]] = new[:, [1, 3]].clip(0, height)
    keep = box_candidates(box1=targets[:, 1:5].T * scale, box2=new.T, area_thr=0.10)
    targets = targets[keep]
    targets[:, 1:5] = new[keep]
    return targets


def hsv_luts(r):
    """augmentations.py:76-79."""
    x = np.arange(0, 256, dtype=r.dtype)
    return ((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8), np.clip(x * r[2], 0, 255).astype(np.uint8)


def letterbox_sample(images, labels, d, s, hyp=None):
    """One training sample of the NON-mosaic branch with augment = True, rect = False (dataloaders.py:710-733): load_image, letterbox to s x s
    (utils/augmentations.py:85-115, auto=False, scaleup=True: the longest side already is s, so only the 114 border is added), labels to
    pixels with the FLOAT pad, random_perspective with border (0, 0), then the common tail.  Same return value as mosaic_sample."""
    i = d["indices"][0]
    im, _, (h, w) = load_image(images[i], s)
    r = min(s / h, s / w)                                   # augmentations.py:93 (scaleup=True)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = (s - nw) / 2, (s - nh) / 2                     # :100-106 (auto=False, scaleFill=False)
    if (w, h) != (nw, nh):
        im = tp.cv2_resize(im, (nw, nh), interpolation=1)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    img = tp.cv2_copy_make_border(im, top, bottom, left, right, 0, value=(114, 114, 114))
    lab = labels[i].copy()
    if lab.size:
        lab[:, 1:] = tp.xywhn2xyxy(lab[:, 1:], r * w, r * h, padw=dw, padh=dh)
    M, width, height = perspective_matrix(d, img.shape[:2], (0, 0))
    if (M != np.eye(3)).any():
        img = tp.cv2_warp_affine(img, M[:2], (width, height), borderValue=(114, 114, 114))
    lab = warp_boxes(lab, M, width, height, d["scale"])
    return _tail(img, lab, d, hyp)


def sample(images, labels, d, s, hyp=None):
    """dataloaders.py:701: the branch the mosaic gate chose."""
    return mosaic_sample(images, labels, d, s, hyp) if d.get("mosaic", True) else letterbox_sample(images, labels, d, s, hyp)


def mosaic_sample(images, labels, d, s, hyp=None, _partner=False):
    """One training sample.  images: list of HWC uint8 BGR arrays; labels: list of (k, 5) float arrays [cls, xc, yc, w, h] normalised;
    d: draws (reference_draws).  Returns (img (3, s, s) uint8 RGB CHW, labels_out (nl, 6) float32 [0, cls, xc, yc, w, h])."""
    tiles_im, shapes = [], []
    for i in d["indices"]:
        im, _, hw = load_image(images[i], s)
        tiles_im.append(im)
        shapes.append(hw)
    rects = mosaic_tiles(shapes, d["yc"], d["xc"], s)
    img4 = np.full((s * 2, s * 2, 3), 114, dtype=np.uint8)
    labels4 = []
    for im, (h, w), i, (x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b) in zip(tiles_im, shapes, d["indices"], rects):
        img4[y1a:y2a, x1a:x2a] = im[y1b:y2b, x1b:x2b]
        lb = labels[i].copy()
        if lb.size:
            lb[:, 1:] = tp.xywhn2xyxy(lb[:, 1:], w, h, x1a - x1b, y1a - y1b)
        labels4.append(lb)
    labels4 = np.concatenate(labels4, 0)
    np.clip(labels4[:, 1:], 0, 2 * s, out=labels4[:, 1:])
    border = (-s // 2, -s // 2)
    M, width, height = perspective_matrix(d, img4.shape[:2], border)
    img = tp.cv2_warp_affine(img4, M[:2], (width, height), borderValue=(114, 114, 114))
    lab = warp_boxes(labels4, M, width, height, d["scale"])
    if d.get("mixup") and not _partner:   # utils/augmentations.py:225-233 on the two warped mosaics, before the common tail
        img2, lab2 = mosaic_sample(images, labels, d["partner"], s, hyp, _partner=True)
        img = (img * d["mix_r"] + img2 * (1 - d["mix_r"])).astype(np.uint8)
        lab = np.concatenate((lab, lab2), 0)
    if _partner:
        return img, lab
    return _tail(img, lab, d, hyp)


def _tail(img, lab, d, hyp):
    """dataloaders.py:735-762, common to both branches: normalised clipped xywh, HSV, flips, CHW / RGB."""
    nl = len(lab)
    if nl:
        lab[:, 1:5] = tp.xyxy2xywhn(lab[:, 1:5], w=img.shape[1], h=img.shape[0], clip=True, eps=1e-3)
    img = img.copy()
    if hyp is None or hyp["hsv_h"] or hyp["hsv_s"] or hyp["hsv_v"]:
        lh, ls, lv = hsv_luts(d["hsv"])
        hsv = tp.cv2_bgr2hsv(img)
        img = tp.cv2_hsv2bgr(np.stack((lh[hsv[..., 0]], ls[hsv[..., 1]], lv[hsv[..., 2]]), -1))
    if d["flipud"]:
        img = np.flipud(img)
        if nl:
            lab[:, 2] = 1 - lab[:, 2]
    if d["fliplr"]:
        img = np.fliplr(img)
        if nl:
            lab[:, 1] = 1 - lab[:, 1]
    out = np.zeros((nl, 6), dtype=np.float32)
    if nl:
        out[:, 1:] = lab.astype(np.float32)
    return np.ascontiguousarray(img.transpose((2, 0, 1))[::-1]), out


def collate(samples):
    """dataloaders.py:858-863."""
    ims, labs = zip(*samples)
    labs = [lb.copy() for lb in labs]
    for i, lb in enumerate(labs):
        lb[:, 0] = i
    return np.stack(ims, 0), np.concatenate(labs, 0)


def synthetic_dataset(n, seed=0, sizes=((96, 128), (128, 96), (80, 80), (128, 128), (60, 100), (110, 70))):
    """n BGR uint8 images of assorted sizes (structured: detgen.scene) with 1-4 normalised boxes each."""
    from . import detgen

    ims, labs = [], []
    for i in range(n):
        h, w = sizes[i % len(sizes)]
        im = (detgen.scene((1, 3, h, w), seed=seed * 100 + i)[0].transpose(1, 2, 0) * 255).round().astype(np.uint8)
        k = 1 + i % 4
        cls = detgen.integers((k, 1), 0, 80, name="acls", seed=seed * 100 + i).astype(np.float64)
        xy = detgen.uniform((k, 2), 0.2, 0.8, name="axy", seed=seed * 100 + i).astype(np.float64)
        wh = detgen.uniform((k, 2), 0.1, 0.5, name="awh", seed=seed * 100 + i).astype(np.float64)
        ims.append(np.ascontiguousarray(im))
        labs.append(np.concatenate((cls, xy, wh), 1))
    return ims, labs
