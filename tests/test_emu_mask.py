"""CPU: y5_process_mask kernel (yolov5_amd/csrc/mask.hip) on the HIP emulator vs the reference-generated golden masks
(tests/golden/mask.npz) and the oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask.npz"))


def run(protos, coef, boxes, shape, upsample, out_u8, ld_extra=0):
    lib = emu()
    c, mh, mw = protos.shape
    n = coef.shape[0]
    P = aligned(protos.shape, protos.dtype); P[...] = protos
    # coefficients / boxes embedded in wider rows, like det[:, 6:] / det[:, :4] of the NMS output
    ld = 6 + c + ld_extra
    det = aligned((n, ld), np.float32)
    det[:, :4] = boxes
    det[:, 6:6 + c] = coef
    oh, ow = shape if upsample else (mh, mw)
    out = aligned((n, oh, ow), np.uint8 if out_u8 else np.float32, 3)
    rc = lib.y5_process_mask(ptr(P), _lib.Y5_F16 if protos.dtype == np.float16 else _lib.Y5_F32, c, mh, mw,
                             C.c_void_p(det.ctypes.data + 24), ld, ptr(det), ld, n, shape[0], shape[1], int(upsample), ptr(out),
                             _lib.Y5_U8 if out_u8 else _lib.Y5_F32, None)
    assert rc == 0, lib.y5_last_error()
    return out


def inputs():
    protos = detgen.uniform((32, 40, 40), -1.0, 1.0, name="protos", seed=13)
    coef = detgen.uniform((7, 32), -1.0, 1.0, name="coef", seed=13)
    xy1 = detgen.uniform((7, 2), 0, 90, name="bx1", seed=13)
    wh = detgen.uniform((7, 2), 8, 70, name="bwh", seed=13)
    return protos, coef, np.concatenate((xy1, xy1 + wh), 1).astype(np.float32)


@pytest.mark.parametrize("up,key", [(False, "noup"), (True, "up")])
@pytest.mark.parametrize("u8", [False, True])
def test_emu_process_mask_vs_reference_golden(up, key, u8):
    protos, coef, boxes = inputs()
    m = run(protos, coef, boxes, (160, 160), up, u8)
    ref = np.unpackbits(G[f"m_{key}"])[: m.size].reshape(G[f"shape_{key}"]).astype(bool)
    assert set(np.unique(m).tolist()) <= {0, 1}
    assert (m.astype(bool) != ref).mean() < 1e-4  # pixels whose bilinear value sits within rounding of 0.5


def test_emu_process_mask_non_square_and_fp16_protos():
    protos = detgen.uniform((8, 24, 36), -2.0, 2.0, name="pr2", seed=3)
    coef = detgen.uniform((5, 8), -1.0, 1.0, name="cf2", seed=3)
    boxes = np.array([[0, 0, 144, 96], [10.5, 3.2, 70.1, 90.0], [100, 50, 144, 96], [30, 30, 31, 31], [-5, -5, 200, 200]], np.float32)
    for dt in (np.float32, np.float16):
        pr = protos.astype(dt)
        m = run(pr, coef, boxes, (96, 144), True, True, ld_extra=3)
        ref = yo.process_mask(torch.from_numpy(pr.astype(np.float32)), torch.from_numpy(coef), torch.from_numpy(boxes), (96, 144), upsample=True).numpy()
        assert (m.astype(bool) != ref.astype(bool)).mean() < 2e-4


def run_batch(protos, dets, shape, upsample, out_u8):
    """y5_process_mask_batch on the emulator: protos (B, c, mh, mw), dets = list of (n_i, 6 + c) NMS-style rows."""
    lib = emu()
    B, c, mh, mw = protos.shape
    P = aligned(protos.shape, protos.dtype); P[...] = protos
    imgs = (_lib.MaskImg * B)()
    keep = []
    for i, d in enumerate(dets):
        a = aligned(d.shape if d.shape[0] else (1, 6 + c), np.float32)
        if d.shape[0]:
            a[...] = d
        keep.append(a)
        imgs[i].masks_in, imgs[i].boxes, imgs[i].ld_m, imgs[i].ld_b, imgs[i].n = a.ctypes.data + 24, a.ctypes.data, 6 + c, 6 + c, d.shape[0]
    oh, ow = shape if upsample else (mh, mw)
    total = sum(d.shape[0] for d in dets)
    out = aligned((max(total, 1), oh, ow), np.uint8 if out_u8 else np.float32, 3)
    rc = lib.y5_process_mask_batch(ptr(P), _lib.Y5_F16 if protos.dtype == np.float16 else _lib.Y5_F32, B, c, mh, mw, imgs, shape[0], shape[1], int(upsample),
                                   ptr(out), _lib.Y5_U8 if out_u8 else _lib.Y5_F32, None)
    assert rc == 0, lib.y5_last_error()
    return out[:total]


@pytest.mark.parametrize("up", [False, True])
@pytest.mark.parametrize("u8", [False, True])
def test_emu_process_mask_batch_equals_per_image(up, u8):
    """The one-launch form (persistent (instance, strip) items, zero strips / zero columns outside the box) against the per-image kernel, pixel for pixel:
    three images (one without detections), boxes inside / at the border / beyond the image / empty / one low-resolution pixel wide, 96 x 192 output
    (several strips, the last one short), fp16 and fp32 prototypes -- and against the oracle."""
    c, mh, mw = 8, 24, 48
    shape = (96, 192)
    for dt in (np.float32, np.float16):
        protos = detgen.uniform((3, c, mh, mw), -2.0, 2.0, name="prb", seed=4).astype(dt)
        boxes = [np.array([[0, 0, 192, 96], [10.5, 3.2, 70.1, 90.0], [130, 50, 192, 96], [30, 30, 31, 31], [-5, -5, 250, 200], [60, 40, 60, 80], [3.9, 90.2, 8.1, 95.9]], np.float32),
                 np.zeros((0, 4), np.float32),
                 np.array([[70, 10, 180, 30], [0, 60, 20, 96]], np.float32)]
        dets = []
        for i, bx in enumerate(boxes):
            d = np.zeros((len(bx), 6 + c), np.float32)
            d[:, :4] = bx
            d[:, 6:] = detgen.uniform((len(bx), c), -1.0, 1.0, name=f"cfb{i}", seed=4)
            dets.append(d)
        got = run_batch(protos, dets, shape, up, u8)
        exp = np.concatenate([run(protos[i], d[:, 6:], d[:, :4], shape, up, u8) for i, d in enumerate(dets) if len(d)])
        assert got.shape == exp.shape and np.array_equal(got, exp)
        ref = np.concatenate([yo.process_mask(torch.from_numpy(protos[i].astype(np.float32)), torch.from_numpy(d[:, 6:].copy()), torch.from_numpy(d[:, :4].copy()),
                                              shape, upsample=up).numpy() for i, d in enumerate(dets) if len(d)])
        assert (got.astype(bool) != ref.astype(bool)).mean() < 2e-4
