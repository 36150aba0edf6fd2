"""CPU: data-gradient of the convolution = forward implicit-GEMM launches on transformed filters (yolov5_amd/train_ops.py,
y5_conv_desc output placement) on the HIP emulator vs torch autograd's conv input gradient."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight
from yolov5_amd.train_ops import dgrad_subconvs

CASES = [
    # B, H, W, C1, C2, k, s, p, accumulate
    (2, 6, 7, 32, 48, 1, 1, 0, False),
    (1, 9, 8, 32, 32, 3, 1, 1, True),
    (2, 12, 10, 16, 32, 3, 2, 1, False),
    (1, 8, 8, 64, 40, 3, 2, 1, True),
    (1, 11, 9, 16, 16, 3, 2, 1, False),   # odd input size: parity classes of different extent
]


@pytest.mark.parametrize("case", CASES)
def test_emu_dgrad_matches_torch(case):
    B, H, W, C1, C2, k, s, p, acc = case
    lib = emu()
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = torch.from_numpy(detgen.uniform((C2, C1, k, k), -0.5, 0.5, name="dw")).half().float()
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="ddz")).half()
    ldz, ldx = C2 + 8, C1 + 8
    dza = aligned((B, OH, OW, ldz), np.float16, 9.0); dza[..., :C2] = dz.permute(0, 2, 3, 1).numpy()
    dx = aligned((B, H, W, ldx), np.float16, 0.0)
    init = detgen.uniform((B, H, W, C1), -1, 1, name="dx0").astype(np.float16)
    if acc:
        dx[..., :C1] = init
    keep = []
    for sub in dgrad_subconvs(w, (s, s), (p, p), (H, W)):
        assert not sub["empty"]
        wp, bp, K, Kpad, Npad = pack_conv_weight(sub["w"], None, torch.float16)
        wa = aligned(wp.shape, np.float16); wa[...] = wp.numpy()
        ba = aligned(bp.shape, np.float32); ba[...] = bp.numpy()
        keep += [wa, ba]
        dense = s == 1
        d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=OH, W=OW, C1=C2, ldx=ldz, OH=sub["nh"], OW=sub["nw"], C2=C1, ldy=ldx,
                          KH=sub["k"][0], KW=sub["k"][1], SH=1, SW=1, PH=sub["pad"][0], PW=sub["pad"][1], act=0, Kpad=Kpad, Npad=Npad,
                          ldr=ldx if acc else 0, ld2=0, cfg=-1, max_blocks=0, out_mul_h=0 if dense else s, out_mul_w=0 if dense else s,
                          out_off_h=sub["rh"], out_off_w=sub["rw"], out_H=0 if dense else H, out_W=0 if dense else W)
        rc = lib.y5_conv2d_fwd(C.byref(d), ptr(dza), ptr(wa), ptr(ba), ptr(dx) if acc else None, ptr(dx), None, None)
        assert rc == 0, lib.y5_last_error()
    x = torch.zeros((B, C1, H, W), requires_grad=True)
    F.conv2d(x, w, None, s, p).backward(dz.float())
    ref = x.grad.permute(0, 2, 3, 1).numpy()
    if acc:
        ref = ref + init.astype(np.float32)
    np.testing.assert_allclose(dx[..., :C1].astype(np.float32), ref, rtol=2e-2, atol=2e-2)
    assert np.all(dx[..., C1:] == 0)
