"""CPU (emulator): the fused backbone front (csrc/conv_front.h, y5_conv_front_fwd) -- 0.Conv k6 s2 p2 from the NCHW batch + 1.Conv k3 s2 p1 + the
pointwise layer behind it (2.C3.cv1 + cv2), models/yolov5s.yaml:17-19 / models/common.py:74-92,246 -- against torch fp32 on the same fp16 data with
the two intermediates rounded to fp16 exactly where the three-launch form stores them to HBM (stem output, 3x3 output)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight, pack_stem_weight


def _mk(rng, shape, fan_in, gain=2.0):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * (gain / fan_in) ** 0.5)


@pytest.mark.parametrize("B,H,W,c1,c3,split,mb,act2", [
    (1, 64, 64, 64, 64, 32, 0, 1),      # one tile per image: every image border at once (top / left zero rows of the stem patch, input padding)
    (2, 64, 128, 64, 64, 64, 1, 1),     # two tiles per image side by side, one workgroup walks all four tiles (input prefetch across tiles), no split
    (1, 128, 64, 32, 32, 16, 2, 0),     # narrow variant (NT1 = NT2 = 1), two workgroups, no activation behind the 1x1
    (1, 64, 64, 56, 48, 16, 0, 1),      # padded channel counts: 56 real 3x3 channels, 48 real pointwise outputs (partial last block: drained waits)
])
def test_front_matches_torch(B, H, W, c1, c3, split, mb, act2):
    lib = emu()
    rng = np.random.default_rng(B * 1000 + H + W + c1 + c3)
    w0, b0 = _mk(rng, (32, 3, 6, 6), 108), torch.from_numpy(rng.standard_normal(32).astype(np.float32) * 0.3)
    w1, b1 = _mk(rng, (c1, 32, 3, 3), 288), torch.from_numpy(rng.standard_normal(c1).astype(np.float32) * 0.3)
    w2, b2 = _mk(rng, (c3, c1, 1, 1), c1), torch.from_numpy(rng.standard_normal(c3).astype(np.float32) * 0.3)
    w0p, b0p, n0 = pack_stem_weight(w0, b0)
    w1p, b1p, _, K1, N1 = pack_conv_weight(w1, b1, torch.float16)
    w2p, b2p, _, K2, N2 = pack_conv_weight(w2, b2, torch.float16)
    assert n0 == 32
    x = aligned((B, 3, H, W), np.float16)
    x[...] = rng.random(x.shape).astype(np.float16)
    OH, OW = H // 4, W // 4
    ldy, ld2 = split + 8, (c3 - split) + 16
    y = aligned((B, OH, OW, ldy), np.float16, 7)
    y2 = aligned((B, OH, OW, ld2), np.float16, 7) if split < c3 else None
    bufs = [aligned(t.shape, t.numpy().dtype) for t in (w0p, b0p, w1p, b1p, w2p, b2p)]
    for dst, src in zip(bufs, (w0p, b0p, w1p, b1p, w2p, b2p)):
        dst[...] = src.numpy()
    W0, B0, W1, B1, W2, B2 = bufs
    rc = lib.y5_conv_front_fwd(ptr(x), B, H, W, ptr(W0), ptr(B0), 32, ptr(W1), ptr(B1), c1, N1, K1, 1, ptr(W2), ptr(B2), c3, N2, K2, act2, ptr(y), ldy,
                               C.c_void_p(y2.ctypes.data + 16 * 2) if y2 is not None else None, ld2, split, mb, None)
    assert rc == 0, lib.y5_last_error()
    xf = torch.from_numpy(x.astype(np.float32))
    t0 = F.silu(F.conv2d(xf, w0.half().float(), b0, stride=2, padding=2)).half().float()
    t1 = F.silu(F.conv2d(t0, w1.half().float(), b1, stride=2, padding=1)).half().float()
    ref = F.conv2d(t1, w2.half().float(), b2)
    ref = (F.silu(ref) if act2 else ref).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y[..., :split].astype(np.float32), ref[..., :split], rtol=5e-3, atol=5e-3)
    assert np.all(y[..., split:] == 7)
    if y2 is not None:
        np.testing.assert_allclose(y2[..., 16:16 + c3 - split].astype(np.float32), ref[..., split:], rtol=5e-3, atol=5e-3)
        assert np.all(y2[..., :16] == 7) and np.all(y2[..., 16 + c3 - split:] == 7)


def test_front_rejects_bad_shapes():
    lib = emu()
    a = aligned((1 << 16,), np.float16)
    f = aligned((64,), np.float32)
    ok = lambda **kw: lib.y5_conv_front_fwd(ptr(a), 1, kw.get("H", 64), 64, ptr(a), ptr(f), kw.get("c0", 32), ptr(a), ptr(f), 64, 64, 320, 1, ptr(a), ptr(f),  # noqa: E731
                                            kw.get("c3", 64), 64, 64, 1, ptr(a), 64, ptr(a), 64, kw.get("split", 32), 0, None)
    assert ok() == 0, lib.y5_last_error()
    assert ok(H=96) != 0 and ok(c0=16) != 0 and ok(c3=72) != 0 and ok(split=12) != 0


def test_plan_with_fused_front_equals_two_launch_plan_on_yolov5s(monkeypatch):
    """The planner's plan with the fused front (Y5_FUSED_FRONT=1: stem + 1.Conv + 2.C3.cv1+cv2 in one launch, then the graph body from op 4) and its
    plan with the two launches (stem, conv+pw) give the same outputs on the emulator -- same fp16 intermediates, held in LDS / registers instead of HBM."""
    import torch

    from oracle import detgen
    from tests.hipemu.backend import EmuBackend
    from tests.test_emu_model import det_model
    from yolov5_amd.engine import Engine

    m = det_model("yolov5s", 0, True).half()
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=0)).half()
    monkeypatch.setenv("Y5_FUSED_K3PW", "1")
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("Y5_FUSED_FRONT", mode)
        eng = Engine(m, (2, 3, 64, 64), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
        outs[mode] = np.asarray(eng(x)["z"]).astype(np.float32).copy()
        assert any(n.startswith("front:") for n in eng.op_names) == (mode == "1"), eng.op_names
        assert (eng._front is not None) == (mode == "1")
        if mode == "1":
            assert [n for n, c in eng.plan_table() if c == "front"] == [eng.op_names[eng._front]]
            t = eng.time_ops(iters=1)
            assert t[0][0].startswith("front:") and eng.timed_order[1] == 4
    u, v = outs["0"], outs["1"]
    assert np.abs(u - v).max() <= 4e-3 * max(1.0, np.abs(u).max()), np.abs(u - v).max()
