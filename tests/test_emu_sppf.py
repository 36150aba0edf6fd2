"""CPU (emulator): SPPF front half in one launch (csrc/conv_sppf.h, y5_sppf_cv1_pool_fwd: cv1 + three cascaded max pools -> four channel slices of the
concat buffer; models/common.py:318-340) against torch on the same fp16 operands, in both LDS-DMA landing models of the emulator."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd.packing import pack_conv_weight

CASES = [(1, 5, 7, 64, 64, 5, 0), (2, 20, 20, 96, 128, 5, 8), (1, 13, 9, 32, 64, 3, 0), (3, 8, 8, 160, 64, 5, 16), (1, 20, 20, 512, 256, 5, 0)]


def _run(B, H, W, C1, c_, k, ld_extra):
    lib = emu()
    rng = np.random.default_rng(B * 100 + H * 10 + W + C1)
    w = torch.from_numpy(rng.standard_normal((c_, C1, 1, 1)).astype(np.float32) * (2.0 / C1) ** 0.5)
    b = torch.from_numpy(rng.standard_normal(c_).astype(np.float32) * 0.3)
    wp, bp, _, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    ldx, ld = C1 + ld_extra, 4 * c_ + ld_extra
    xbuf = aligned((B, H, W, ldx), np.float16)
    xbuf[...] = rng.standard_normal(xbuf.shape).astype(np.float16)
    x = xbuf[..., ldx - C1:]
    buf = aligned((B, H, W, ld), np.float16, 7)
    Wp, Bp = aligned(wp.shape, np.float16), aligned(bp.shape, np.float32)
    Wp[...] = wp.numpy(); Bp[...] = bp.numpy()
    rc = lib.y5_sppf_cv1_pool_fwd(C.c_void_p(xbuf.ctypes.data + (ldx - C1) * 2), ldx, ptr(Wp), ptr(Bp), Kpad, ptr(buf), ld, B, H, W, C1, c_, k, 1, None)
    assert rc == 0, lib.y5_last_error()
    xf = torch.from_numpy(np.ascontiguousarray(x).astype(np.float32)).permute(0, 3, 1, 2)
    y0 = F.silu(F.conv2d(xf, w.half().float(), b)).half().float()
    ys = [y0]
    for _ in range(3):
        ys.append(F.max_pool2d(ys[-1], k, 1, k // 2))
    ref = torch.cat(ys, 1).permute(0, 2, 3, 1).numpy()
    got = buf[..., :4 * c_].astype(np.float32)
    np.testing.assert_allclose(got[..., :c_], ref[..., :c_], rtol=4e-3, atol=4e-3)
    # the pools are exact on whatever cv1 produced: slices 1..3 against torch's pools of the KERNEL's slice 0
    g0 = torch.from_numpy(got[..., :c_]).permute(0, 3, 1, 2)
    for s in range(1, 4):
        g0 = F.max_pool2d(g0, k, 1, k // 2)
        assert np.array_equal(got[..., s * c_:(s + 1) * c_], g0.permute(0, 2, 3, 1).numpy()), s
    assert np.all(buf[..., 4 * c_:] == 7)


@pytest.mark.parametrize("B,H,W,C1,c_,k,ld_extra,async_dma", [c + ("0",) for c in CASES] + [CASES[i] + ("1",) for i in (1, 3)])
def test_sppf_cv1_pool_matches_torch(B, H, W, C1, c_, k, ld_extra, async_dma):
    if async_dma == "1":
        code = f"import tests.test_emu_sppf as t; t._run({B}, {H}, {W}, {C1}, {c_}, {k}, {ld_extra})"
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, Y5_EMU_ASYNC="1"), capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return
    _run(B, H, W, C1, c_, k, ld_extra)


def test_sppf_cv1_pool_rejects_what_it_cannot_do():
    lib = emu()
    a = aligned((1, 24, 24, 64), np.float16)
    w = aligned((64, 64), np.float16)
    b = aligned((64,), np.float32)
    out = aligned((1, 24, 24, 256), np.float16)
    assert lib.y5_sppf_cv1_pool_fwd(ptr(a), 64, ptr(w), ptr(b), 64, ptr(out), 256, 1, 24, 24, 64, 64, 5, 1, None) != 0      # 576 pixels per image
    assert lib.y5_sppf_cv1_pool_fwd(ptr(a), 64, ptr(w), ptr(b), 64, ptr(out), 256, 1, 8, 8, 64, 32, 5, 1, None) != 0        # c_ not a multiple of 64
    assert lib.y5_sppf_cv1_pool_fwd(ptr(a), 64, ptr(w), ptr(b), 64, ptr(out), 128, 1, 8, 8, 64, 64, 5, 1, None) != 0        # concat buffer too narrow


def test_plan_runs_sppf_front_as_one_launch(monkeypatch):
    """yolov5s 9.SPPF: the plan with cv1 + pools as one launch (default) against cv1 + y5_sppf_pool (Y5_FUSED_SPPF=0) on the emulator."""
    from oracle import detgen
    from tests.hipemu.backend import EmuBackend
    from tests.test_emu_model import det_model
    from yolov5_amd.engine import Engine

    m = det_model("yolov5s", 0, True).half()
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 96), 0.0, 1.0, name="img", seed=0)).half()
    outs, names = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("Y5_FUSED_SPPF", "force" if mode == "1" else mode)   # (force: below the planner's workgroup-count gate at this batch)
        eng = Engine(m, (2, 3, 64, 96), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
        outs[mode] = np.asarray(eng(x)["z"]).astype(np.float32).copy()
        names[mode] = list(eng.op_names)
    assert any(n.startswith("sppf_front:") for n in names["1"]) and "sppf_pool" not in names["1"]
    assert "sppf_pool" in names["0"] and not any(n.startswith("sppf_front:") for n in names["0"])
    u, v = outs["0"], outs["1"]
    assert np.abs(u - v).max() <= 4e-3 * max(1.0, np.abs(u).max()), np.abs(u - v).max()
