"""CPU (emulator): `Conv(32, 64, 3, 2)` + the pointwise convolution behind it as one launch (csrc/conv_k3.h PW2, y5_conv_k3pw_fwd;
models/yolo.py walking 1.Conv -> 2.C3 whose cv1 + cv2 GEMM, common.py:246, is the Conv's only reader) against torch fp32 on the same
fp16 data -- the intermediate is stored as fp16 (LDS) exactly like the two-launch form stores it to HBM -- and the planner's fused plan
against its own unfused plan on yolov5s."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight


@pytest.mark.parametrize("B,H,W,c2,c3,split,ldx,cfg,mb,act2", [
    (1, 8, 16, 64, 64, 32, 32, 34, 0, 1),       # one wave tile row, split like C3's cv1 | cv2
    (2, 16, 32, 64, 64, 64, 40, 31, 1, 1),      # everything to y, input a slice of a wider buffer, one workgroup walks all tiles (2-stage ring)
    (1, 24, 48, 64, 48, 16, 32, 34, 2, 0),      # 48 real output channels (padded to 64), no activation behind the 1x1, 3-stage ring steady state
    (3, 8, 16, 56, 64, 32, 32, -1, 0, 1),       # 3x3 with 56 real channels (padded filter rows / columns are zeros)
    (2, 16, 48, 64, 64, 32, 32, 81, 1, 1),      # eight waves per workgroup, one stage each
])
def test_k3pw_matches_torch(B, H, W, c2, c3, split, ldx, cfg, mb, act2):
    lib = emu()
    rng = np.random.default_rng(B * 100 + H + c3)
    C1 = 32
    w1 = torch.from_numpy(rng.standard_normal((c2, C1, 3, 3)).astype(np.float32) * (2.0 / (9 * C1)) ** 0.5)
    b1 = torch.from_numpy(rng.standard_normal(c2).astype(np.float32) * 0.3)
    w2 = torch.from_numpy(rng.standard_normal((c3, c2, 1, 1)).astype(np.float32) * (2.0 / c2) ** 0.5)
    b2 = torch.from_numpy(rng.standard_normal(c3).astype(np.float32) * 0.3)
    w1p, b1p, _, K1, N1 = pack_conv_weight(w1, b1, torch.float16)
    w2p, b2p, _, K2, N2 = pack_conv_weight(w2, b2, torch.float16)
    assert N1 == 64 and N2 == 64
    xbuf = aligned((B, H, W, ldx), np.float16)
    xbuf[...] = rng.standard_normal(xbuf.shape).astype(np.float16)
    OH, OW = H // 2, W // 2
    ldy, ld2 = split + 8, (c3 - split) + 16
    y = aligned((B, OH, OW, ldy), np.float16, 7)
    y2 = aligned((B, OH, OW, ld2), np.float16, 7) if split < c3 else None
    W1, B1, W2, B2 = (aligned(t.shape, t.numpy().dtype) for t in (w1p, b1p, w2p, b2p))
    for dst, src in ((W1, w1p), (B1, b1p), (W2, w2p), (B2, b2p)):
        dst[...] = src.numpy()
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=ldx, OH=OH, OW=OW, C2=c2, ldy=64, KH=3, KW=3, SH=2, SW=2, PH=1, PW=1, act=1,
                      Kpad=K1, Npad=N1, ldr=0, ld2=0, cfg=cfg, max_blocks=mb)
    xoff = (ldx - C1) * 2
    rc = lib.y5_conv_k3pw_fwd(C.byref(d), C.c_void_p(xbuf.ctypes.data + xoff), ptr(W1), ptr(B1), ptr(W2), ptr(B2), c3, N2, K2, act2, ptr(y), ldy,
                              C.c_void_p(y2.ctypes.data + 16 * 2) if y2 is not None else None, ld2, split, None)
    assert rc == 0, lib.y5_last_error()
    xf = torch.from_numpy(np.ascontiguousarray(xbuf[..., ldx - C1:]).astype(np.float32)).permute(0, 3, 1, 2)
    t = F.silu(F.conv2d(xf, w1.half().float(), b1, stride=2, padding=1)).half().float()
    ref = F.conv2d(t, w2.half().float(), b2)
    ref = (F.silu(ref) if act2 else ref).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y[..., :split].astype(np.float32), ref[..., :split], rtol=4e-3, atol=4e-3)
    assert np.all(y[..., split:] == 7)
    if y2 is not None:
        np.testing.assert_allclose(y2[..., 16:16 + c3 - split].astype(np.float32), ref[..., split:], rtol=4e-3, atol=4e-3)
        assert np.all(y2[..., :16] == 7) and np.all(y2[..., 16 + c3 - split:] == 7)


def test_k3pw_rejects_bad_shapes():
    lib = emu()
    a = aligned((4096,), np.float16)
    f = aligned((64,), np.float32)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=1, H=8, W=16, C1=32, ldx=32, OH=4, OW=8, C2=64, ldy=64, KH=3, KW=3, SH=2, SW=2, PH=1, PW=1, act=1,
                      Kpad=320, Npad=64, cfg=-1)
    ok = lambda **kw: lib.y5_conv_k3pw_fwd(C.byref(d), ptr(a), ptr(a), ptr(f), ptr(a), ptr(f), kw.get("c3", 64), 64, 64, 1, ptr(a), 64, ptr(a), 64,  # noqa: E731
                                           kw.get("split", 32), None)
    assert ok() == 0, lib.y5_last_error()
    assert ok(c3=72) != 0 and ok(split=12) != 0
    d.SH = 1
    assert ok() != 0
    d.SH = 2
    d.cfg = 8
    assert ok() != 0


def test_plan_fuses_conv1_into_c3_on_yolov5s(monkeypatch):
    """The planner's fused plan (Y5_FUSED_K3PW=1) and its unfused plan give the same outputs on the emulator: the fp16 intermediate is the same
    tensor in both (LDS instead of HBM) and the second GEMM accumulates the same k order in one fp32 chain."""
    from oracle import detgen
    from tests.hipemu.backend import EmuBackend
    from tests.test_emu_model import det_model
    from yolov5_amd.engine import Engine

    m = det_model("yolov5s", 0, True).half()
    x = torch.from_numpy(detgen.uniform((1, 3, 64, 64), 0.0, 1.0, name="img", seed=0)).half()
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("Y5_FUSED_K3PW", mode)
        eng = Engine(m, (1, 3, 64, 64), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
        outs[mode] = np.asarray(eng(x)["z"]).astype(np.float32).copy()
        assert any(n.startswith("conv+pw:") for n in eng.op_names) == (mode == "1"), eng.op_names
    u, v = outs["0"], outs["1"]
    assert np.abs(u - v).max() <= 4e-3 * max(1.0, np.abs(u).max()), np.abs(u - v).max()
