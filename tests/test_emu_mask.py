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
