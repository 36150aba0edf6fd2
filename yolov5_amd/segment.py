"""Segmentation post-processing with the reference's names (utils/segment/general.py): `process_mask` :25-51 and
`crop_mask` :10-22 run as one HIP kernel per image (y5_process_mask); `process_mask_batch` does the per-image loop of segment/predict.py:161-172
for a whole batch in ONE launch (y5_process_mask_batch)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def process_mask(protos, masks_in, bboxes, shape, upsample=False, out_dtype=torch.float32):
    """protos (c, mh, mw) GPU f16|f32; masks_in (n, c); bboxes (n, 4) xyxy in input pixels; shape (ih, iw).
    Returns (n, ih, iw) if upsample else (n, mh, mw), values 0/1.  `out_dtype`: torch.float32 (what the reference's
    `masks.gt_(0.5)` returns) or torch.bool / torch.uint8 (4x fewer HBM bytes).  masks_in / bboxes may be column
    views of the NMS output rows (`det[:, 6:]`, `det[:, :4]`): they are read in place through their row stride."""
    if not _lib.accepts(protos):
        raise RuntimeError("yolov5_amd.process_mask needs GPU tensors (no CPU path)")
    lib = _lib.lib()
    c, mh, mw = protos.shape
    ih, iw = int(shape[0]), int(shape[1])
    n = int(masks_in.shape[0])
    if protos.dtype not in (torch.float16, torch.float32):
        protos = protos.float()
    protos = protos.contiguous()

    def rows(t, width):
        if t.dtype != torch.float32 or t.stride(-1) != 1 or (t.shape[0] > 1 and t.stride(0) < width):
            t = t.float().contiguous()
        return t, (t.stride(0) if t.shape[0] > 1 else width)

    masks_in, ld_m = rows(masks_in, c)
    bboxes, ld_b = rows(bboxes, 4)
    oh, ow = (ih, iw) if upsample else (mh, mw)
    u8 = out_dtype in (torch.bool, torch.uint8)
    if not u8 and out_dtype != torch.float32:
        raise TypeError("process_mask: out_dtype must be float32, uint8 or bool")
    out = torch.empty((n, oh, ow), dtype=torch.uint8 if u8 else torch.float32, device=protos.device)
    rc = lib.y5_process_mask(C.c_void_p(protos.data_ptr()), _lib.Y5_F16 if protos.dtype == torch.float16 else _lib.Y5_F32,
                             c, mh, mw, C.c_void_p(masks_in.data_ptr()), ld_m, C.c_void_p(bboxes.data_ptr()), ld_b, n, ih, iw,
                             1 if upsample else 0, C.c_void_p(out.data_ptr()), _lib.Y5_U8 if u8 else _lib.Y5_F32,
                             _lib.stream(protos.device))
    _lib.check(rc, lib)
    return out.view(torch.bool) if out_dtype == torch.bool else out


def _rows(t, width):
    if t.dtype != torch.float32 or t.stride(-1) != 1 or (t.shape[0] > 1 and t.stride(0) < width):
        t = t.float().contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else width)


def process_mask_batch(protos, dets, shape, upsample=False, out_dtype=torch.float32):
    """`[process_mask(protos[i], det[:, 6:], det[:, :4], shape, upsample) for i, det in enumerate(dets)]` (segment/predict.py:161-172) as ONE launch.
    protos (B, c, mh, mw) GPU f16|f32; dets: the per-image NMS results, (n_i, 6 + c) fp32 rows [x1, y1, x2, y2, conf, cls, coefficients] -- read in place
    (they are views of the padded NMS buffer).  Returns the list of per-image masks (n_i, oh, ow) -- views of one buffer -- with values 0/1 in `out_dtype`
    (torch.float32: the reference's `masks.gt_(0.5)`; torch.bool / torch.uint8: 4x fewer HBM bytes).  Shapes the batched kernel does not take (output width not
    a multiple of 16 bytes) fall back to the per-image kernel."""
    if not _lib.accepts(protos):
        raise RuntimeError("yolov5_amd.process_mask_batch needs GPU tensors (no CPU path)")
    lib = _lib.lib()
    B, c, mh, mw = protos.shape
    if len(dets) != B:
        raise ValueError(f"process_mask_batch: {len(dets)} detection tensors for {B} prototype sets")
    ih, iw = int(shape[0]), int(shape[1])
    oh, ow = (ih, iw) if upsample else (mh, mw)
    u8 = out_dtype in (torch.bool, torch.uint8)
    if not u8 and out_dtype != torch.float32:
        raise TypeError("process_mask_batch: out_dtype must be float32, uint8 or bool")
    if protos.dtype not in (torch.float16, torch.float32):
        protos = protos.float()
    protos = protos.contiguous()
    if ow % (16 if u8 else 4):
        return [process_mask(protos[i], d[:, 6:], d[:, :4], shape, upsample, out_dtype) for i, d in enumerate(dets)]
    imgs = (_lib.MaskImg * B)()
    keep = []   # tensors the descriptors point into
    ns = []
    for i, d in enumerate(dets):
        n = int(d.shape[0])
        ns.append(n)
        if n == 0:
            imgs[i].n = 0
            continue
        if d.shape[1] != 6 + c:
            raise ValueError(f"process_mask_batch: detection rows have {d.shape[1]} columns, expected 6 + {c}")
        d, ld = _rows(d, 6 + c)
        keep.append(d)
        imgs[i].masks_in, imgs[i].boxes, imgs[i].ld_m, imgs[i].ld_b, imgs[i].n = d.data_ptr() + 24, d.data_ptr(), ld, ld, n
    total = sum(ns)
    out = torch.empty((total, oh, ow), dtype=torch.uint8 if u8 else torch.float32, device=protos.device)
    if total:
        rc = lib.y5_process_mask_batch(C.c_void_p(protos.data_ptr()), _lib.Y5_F16 if protos.dtype == torch.float16 else _lib.Y5_F32, B, c, mh, mw, imgs,
                                       ih, iw, 1 if upsample else 0, C.c_void_p(out.data_ptr()), _lib.Y5_U8 if u8 else _lib.Y5_F32, _lib.stream(protos.device))
        _lib.check(rc, lib)
    if out_dtype == torch.bool:
        out = out.view(torch.bool)
    return list(out.split(ns))
