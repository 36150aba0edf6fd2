"""Image pre-processing on the device, reference names: utils/augmentations.py `letterbox`.

    letterbox_geometry(shape, ...)    the arithmetic of utils/augmentations.py:85-115 (ratio, new_unpad, border) -- host
    letterbox(im, ...)                same signature / return value as the reference for ONE image (device uint8 HWC)
    letterbox_batch(ims, ...)         letterbox + HWC->CHW (+BGR->RGB) + `.half() / 255` for a batch in ONE launch
                                      (`csrc/preprocess.hip` -> y5_letterbox_batch): the tensor the model consumes and
                                      the per-image `shapes` entries that scale_boxes / metrics.match_batch need.
The resize restates cv2.INTER_LINEAR on 8-bit images; opencv-python is a third-party dependency that is absent here, so
that layer is parity-unpinned -- the geometry is pinned against the reference's own `letterbox`.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scaleFill=False, scaleup=True, stride=32):
    """utils/augmentations.py:87-112.  shape = (h0, w0) -> dict(new_unpad=(w, h), ratio=(rw, rh), pad=(dw, dh) [per side, float],
    top, bottom, left, right, out_shape=(H, W))."""
    h0, w0 = int(shape[0]), int(shape[1])
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / h0, new_shape[1] / w0)
    if not scaleup:
        r = min(r, 1.0)
    ratio = (r, r)
    new_unpad = (round(w0 * r), round(h0 * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride  # np.mod on non-negative ints / floats
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = (new_shape[1] / w0, new_shape[0] / h0)
    dw /= 2
    dh /= 2
    top, bottom = round(dh - 0.1), round(dh + 0.1)
    left, right = round(dw - 0.1), round(dw + 0.1)
    return dict(new_unpad=new_unpad, ratio=ratio, pad=(dw, dh), top=top, bottom=bottom, left=left, right=right,
                out_shape=(new_unpad[1] + top + bottom, new_unpad[0] + left + right))


def _pad_value(color):
    c = tuple(color) if hasattr(color, "__len__") else (color,) * 3
    if len(set(int(v) for v in c)) != 1:
        raise NotImplementedError("letterbox: the border colour must be the same in all channels")
    return int(c[0])


def _check_image(im):
    if not (torch.is_tensor(im) and _lib.accepts(im)):
        raise RuntimeError("yolov5_amd.augmentations.letterbox needs uint8 HWC GPU tensors (no CPU path)")
    if im.dtype != torch.uint8 or im.ndim != 3 or im.shape[2] != 3:
        raise ValueError(f"letterbox: expected uint8 (h, w, 3), got {im.dtype} {tuple(im.shape)}")
    return im if im.is_contiguous() else im.contiguous()


def _launch(ims, geos, H, W, pad, swap_rb, out, chw, div255):
    jobs = (_lib.LetterboxJob * len(ims))()
    for j, im, g in zip(jobs, ims, geos):
        j.src, j.h0, j.w0, j.stride = im.data_ptr(), im.shape[0], im.shape[1], im.stride(0)
        j.nw, j.nh, j.top, j.left = g["new_unpad"][0], g["new_unpad"][1], g["top"], g["left"]
    dev = out.device
    table = torch.frombuffer(bytearray(jobs), dtype=torch.uint8).to(dev, non_blocking=False)
    code = {torch.uint8: _lib.Y5_U8, torch.float16: _lib.Y5_F16, torch.float32: _lib.Y5_F32}[out.dtype]
    lib = _lib.lib()
    rc = lib.y5_letterbox_batch(C.c_void_p(table.data_ptr()), len(ims), H, W, pad, int(swap_rb), C.c_void_p(out.data_ptr()), code, int(chw),
                                int(div255), _lib.stream(dev))
    _lib.check(rc, lib)
    return out  # `table` may be released here: the caching allocator re-uses it in stream order, after the launch


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    """utils/augmentations.py:85-115 on a device uint8 (h, w, 3) image -> (uint8 (H, W, 3) on the device, ratio, (dw, dh))."""
    im = _check_image(im)
    g = letterbox_geometry(im.shape[:2], new_shape, auto, scaleFill, scaleup, stride)
    H, W = g["out_shape"]
    out = torch.empty((1, H, W, 3), dtype=torch.uint8, device=im.device)
    _launch([im], [g], H, W, _pad_value(color), False, out, False, False)
    return out[0], g["ratio"], g["pad"]


def letterbox_batch(ims, new_shape=(640, 640), color=(114, 114, 114), auto=False, scaleFill=False, scaleup=True, stride=32,
                    dtype=torch.float16, swap_rb=False, normalize=True):
    """Letterbox every image of `ims` (device uint8 HWC tensors of any sizes) to one common shape and produce the model input:
    (B, 3, H, W) `dtype`, channels reversed when swap_rb (cv2 BGR input, detect.py:205), divided by 255 when normalize
    (detect.py:208-209).  dtype=torch.uint8 gives the un-normalised CHW batch of models/common.py:925.
    Returns (x, shapes) with shapes[i] = ((h0, w0), ((rw, rh), (dw, dh))) -- the dataloader's `shapes` entry (utils/dataloaders.py
    `shapes = (h0, w0), ((h / h0, w / w0), pad)`), input of scale_boxes(ratio_pad=...) and metrics.match_batch."""
    ims = [_check_image(im) for im in ims]
    if not ims:
        raise ValueError("letterbox_batch: empty batch")
    geos = [letterbox_geometry(im.shape[:2], new_shape, auto, scaleFill, scaleup, stride) for im in ims]
    H, W = geos[0]["out_shape"]
    if any(g["out_shape"] != (H, W) for g in geos):
        raise ValueError("letterbox_batch: images letterbox to different shapes (auto=True): " + str(sorted({g['out_shape'] for g in geos})))
    out = torch.empty((len(ims), 3, H, W), dtype=dtype, device=ims[0].device)
    _launch(ims, geos, H, W, _pad_value(color), swap_rb, out, True, normalize and dtype != torch.uint8)
    shapes = [((im.shape[0], im.shape[1]), (g["ratio"], g["pad"])) for im, g in zip(ims, geos)]
    return out, shapes
