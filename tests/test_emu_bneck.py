"""CPU (emulator): the fused Bottleneck kernel (csrc/conv_bneck.h, y5_bottleneck_fwd: 1x1 -> SiLU -> 3x3 -> SiLU -> [+ x] with the
intermediate in LDS; models/common.py:164-181) against torch fp32 on the same fp16 data, and the split store of y5_conv2d_fwd
(C3's cv1 + cv2 as one GEMM whose halves land in two buffers, models/common.py:246)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight


def _ref_bneck(x, w1, b1, w2, b2, add):
    """x (B,H,W,C) fp16 values; t is stored in fp16 by the kernel (LDS), accumulation fp32."""
    xf = torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2)
    t = F.silu(F.conv2d(xf, w1.half().float(), b1)).half().float()
    y = F.silu(F.conv2d(t, w2.half().float(), b2, padding=1)).half().float()
    if add:
        y = (y + xf).half().float()
    return y.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("Cc,B,H,W,add,ldx,ldy,mb", [(32, 2, 8, 16, True, 32, 32, 0), (32, 1, 12, 8, False, 64, 40, 0), (64, 1, 8, 8, True, 128, 64, 0),
                                                      (64, 2, 4, 24, True, 64, 72, 0), (32, 3, 4, 8, True, 32, 32, 0),
                                                      # many tiles per wave (grid capped at 1-2 workgroups) for every ring depth: ramp, steady state, drain
                                                      (32, 1, 24, 32, True, 32, 32, 1 | (1 << 16)), (32, 1, 24, 32, True, 64, 32, 1 | (2 << 16)),
                                                      (32, 2, 16, 40, False, 32, 32, 2 | (3 << 16)), (64, 1, 16, 32, True, 64, 64, 1 | (1 << 16)),
                                                      (64, 2, 16, 32, True, 64, 64, 1), (64, 3, 8, 40, False, 128, 64, 2)])   # C = 64 default: eight waves, t aliased onto the stage; 3-4 tiles per wave
def test_fused_bottleneck_matches_torch(Cc, B, H, W, add, ldx, ldy, mb):
    lib = emu()
    rng = np.random.default_rng(Cc + H + W)
    w1 = torch.from_numpy(rng.standard_normal((Cc, Cc, 1, 1)).astype(np.float32) * (2.0 / Cc) ** 0.5)
    w2 = torch.from_numpy(rng.standard_normal((Cc, Cc, 3, 3)).astype(np.float32) * (2.0 / (9 * Cc)) ** 0.5)
    b1 = torch.from_numpy(rng.standard_normal(Cc).astype(np.float32) * 0.3)
    b2 = torch.from_numpy(rng.standard_normal(Cc).astype(np.float32) * 0.3)
    w1p, b1p, _, K1, _ = pack_conv_weight(w1, b1, torch.float16)
    w2p, b2p, _, K2, _ = pack_conv_weight(w2, b2, torch.float16)
    xbuf = aligned((B, H, W, ldx), np.float16)
    xbuf[...] = rng.standard_normal(xbuf.shape).astype(np.float16)
    x = xbuf[..., ldx - Cc:]                                        # a channel slice at an offset, like a concat buffer's
    ybuf = aligned((B, H, W, ldy), np.float16, 7)
    W1, B1, W2, B2 = (aligned(t.shape, t.numpy().dtype) for t in (w1p, b1p, w2p, b2p))
    for dst, src in ((W1, w1p), (B1, b1p), (W2, w2p), (B2, b2p)):
        dst[...] = src.numpy()
    xoff = (ldx - Cc) * 2
    rc = lib.y5_bottleneck_fwd(C.c_void_p(xbuf.ctypes.data + xoff), ldx, ptr(W1), ptr(B1), K1, ptr(W2), ptr(B2), K2, ptr(ybuf), ldy, B, H, W, Cc,
                               int(add), mb, None)
    assert rc == 0, lib.y5_last_error()
    ref = _ref_bneck(np.ascontiguousarray(x), w1, b1, w2, b2, add)
    got = ybuf[..., :Cc].astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=4e-3, atol=4e-3)
    assert np.all(ybuf[..., Cc:] == 7)                               # nothing written outside the slice


# c_ = 128 (conv_h3b.h): GEMM-1 phase + halo-resident 3x3.  Shapes: one tile per image; images cut into several row / column tiles with ragged
# edges (the in-image test of the t store); several tiles per workgroup (next-tile halo prefetch into dead planes, ring re-prologue); 5-stage ring.
# Each case runs with LDS-DMA landing at issue (write-after-read hazards) and at the covering vmcnt wait (counted-wait bookkeeping).
H3B_CASES = [(1, 6, 7, True, 128, 128, 0), (2, 9, 20, False, 256, 136, 0), (1, 23, 40, True, 128, 128, 2), (3, 12, 10, True, 128, 256, 1),
             (1, 40, 40, True, 128, 128, 3), (2, 17, 33, False, 128, 128, 2 | (5 << 16)), (5, 5, 5, True, 128, 128, 1 | (5 << 16)),
             (1, 23, 40, True, 128, 128, 2 | (4 << 16)), (3, 12, 10, False, 128, 256, 1 | (4 << 16)),
             (1, 23, 40, True, 128, 128, 2 | (18 << 16)), (3, 12, 10, False, 128, 256, 1 | (18 << 16)), (2, 17, 33, True, 128, 128, 2 | (18 << 16))]


# (every case with LDS-DMA landing at issue; the worst-case landing model -- a child process each -- on one case per ring form)
@pytest.mark.parametrize("B,H,W,add,ldx,ldy,mb,async_dma", [c + ("0",) for c in H3B_CASES] + [H3B_CASES[i] + ("1",) for i in (2, 5, 7, 9)])
def test_fused_bottleneck_c128_matches_torch(B, H, W, add, ldx, ldy, mb, async_dma):
    import subprocess
    import sys

    if async_dma == "1":   # the DMA model is latched per process (getenv once): run the same case in a child with the switch set
        code = f"import tests.test_emu_bneck as t; t._run_c128({B}, {H}, {W}, {add}, {ldx}, {ldy}, {mb})"
        import os
        env = dict(os.environ, Y5_EMU_ASYNC="1")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return
    _run_c128(B, H, W, add, ldx, ldy, mb)


def _run_c128(B, H, W, add, ldx, ldy, mb):
    lib = emu()
    Cc = 128
    rng = np.random.default_rng(B * 1000 + H * 10 + W)
    w1 = torch.from_numpy(rng.standard_normal((Cc, Cc, 1, 1)).astype(np.float32) * (2.0 / Cc) ** 0.5)
    w2 = torch.from_numpy(rng.standard_normal((Cc, Cc, 3, 3)).astype(np.float32) * (2.0 / (9 * Cc)) ** 0.5)
    b1 = torch.from_numpy(rng.standard_normal(Cc).astype(np.float32) * 0.3)
    b2 = torch.from_numpy(rng.standard_normal(Cc).astype(np.float32) * 0.3)
    w1p, b1p, _, K1, _ = pack_conv_weight(w1, b1, torch.float16)
    w2p, b2p, _, K2, _ = pack_conv_weight(w2, b2, torch.float16)
    xbuf = aligned((B, H, W, ldx), np.float16)
    xbuf[...] = rng.standard_normal(xbuf.shape).astype(np.float16)
    x = xbuf[..., ldx - Cc:]
    ybuf = aligned((B, H, W, ldy), np.float16, 7)
    W1, B1, W2, B2 = (aligned(t.shape, t.numpy().dtype) for t in (w1p, b1p, w2p, b2p))
    for dst, src in ((W1, w1p), (B1, b1p), (W2, w2p), (B2, b2p)):
        dst[...] = src.numpy()
    rc = lib.y5_bottleneck_fwd(C.c_void_p(xbuf.ctypes.data + (ldx - Cc) * 2), ldx, ptr(W1), ptr(B1), K1, ptr(W2), ptr(B2), K2, ptr(ybuf), ldy, B, H, W, Cc,
                               int(add), mb, None)
    assert rc == 0, lib.y5_last_error()
    ref = _ref_bneck(np.ascontiguousarray(x), w1, b1, w2, b2, add)
    got = ybuf[..., :Cc].astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=4e-3, atol=4e-3)
    assert np.all(ybuf[..., Cc:] == 7)


def test_fused_bottleneck_rejects_overlap_and_bad_shapes():
    lib = emu()
    buf = aligned((1, 8, 8, 64), np.float16)
    w = aligned((32, 320), np.float16)
    b = aligned((32,), np.float32)
    args = lambda x, y, ldx=64, ldy=64, Cc=32, H=8, W=8: (x, ldx, ptr(w), ptr(b), 64, ptr(w), ptr(b), 320, y, ldy, 1, H, W, Cc, 1, 0, None)  # noqa: E731
    assert lib.y5_bottleneck_fwd(*args(ptr(buf), ptr(buf))) != 0                                   # in place: neighbours' halos would be clobbered
    assert lib.y5_bottleneck_fwd(*args(ptr(buf), C.c_void_p(buf.ctypes.data + 64))) == 0          # disjoint channel slices of one buffer are fine
    assert lib.y5_bottleneck_fwd(*args(ptr(buf), C.c_void_p(buf.ctypes.data + 64), Cc=48)) != 0
    assert lib.y5_bottleneck_fwd(*args(ptr(buf), C.c_void_p(buf.ctypes.data + 64), H=6)) != 0


@pytest.mark.parametrize("cfg", [-1, 2, 17, 93, 95, 96])
def test_conv_split_store(cfg):
    """1x1 64 -> 64 conv whose channels [0, 32) go to one buffer and [32, 64) to a slice of another (desc.split_n)."""
    lib = emu()
    rng = np.random.default_rng(5)
    B, H, W, C1, C2 = 2, 8, 8, 64, 64
    w = torch.from_numpy(rng.standard_normal((C2, C1, 1, 1)).astype(np.float32) * 0.2)
    b = torch.from_numpy(rng.standard_normal(C2).astype(np.float32) * 0.2)
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    x = aligned((B, H, W, C1), np.float16)
    x[...] = rng.standard_normal(x.shape).astype(np.float16)
    lo = aligned((B, H, W, 32), np.float16, 3)
    hi = aligned((B, H, W, 96), np.float16, 3)
    Wp, Bp = aligned(wp.shape, np.float16), aligned(bp.shape, np.float32)
    Wp[...] = wp.numpy(); Bp[...] = bp.numpy()
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=C1, OH=H, OW=W, C2=C2, ldy=32, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0, act=1,
                      Kpad=Kpad, Npad=Npad, ldr=0, ld2=96, cfg=cfg, max_blocks=0, split_n=32)
    rc = lib.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(Wp), ptr(Bp), None, ptr(lo), C.c_void_p(hi.ctypes.data + 64 * 2), None)
    if cfg == 17 and rc != 0:
        pytest.skip("pointwise configuration 17 does not take this shape")
    assert rc == 0, lib.y5_last_error()
    ref = F.silu(F.conv2d(torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2), w.half().float(), b)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(lo.astype(np.float32), ref[..., :32], rtol=3e-3, atol=3e-3)
    np.testing.assert_allclose(hi[..., 64:].astype(np.float32), ref[..., 32:], rtol=3e-3, atol=3e-3)
    assert np.all(hi[..., :64] == 3)


@pytest.mark.parametrize("B,H,W,add,c3,ldx,ld2,ldo,act3,mb", [(1, 8, 16, True, 64, 32, 64, 64, 1, 0), (2, 12, 8, False, 64, 64, 96, 72, 1, 0),
                                                             (1, 24, 32, True, 48, 32, 64, 48, 0, 1), (3, 4, 8, True, 64, 32, 32, 64, 1, 2)])
def test_fused_bottleneck_cv3_matches_torch(B, H, W, add, c3, ldx, ld2, ldo, act3, mb):
    """y5_bottleneck_cv3_fwd: the last Bottleneck of a C3 + the C3's cv3 (models/common.py:246) in one launch; the Bottleneck's result is rounded to
    fp16 (LDS) exactly as the two-launch form stores it, cat order = (m output, cv2 output)."""
    lib = emu()
    Cc = 32
    rng = np.random.default_rng(H * W + c3)
    w1 = torch.from_numpy(rng.standard_normal((Cc, Cc, 1, 1)).astype(np.float32) * (2.0 / Cc) ** 0.5)
    w2 = torch.from_numpy(rng.standard_normal((Cc, Cc, 3, 3)).astype(np.float32) * (2.0 / (9 * Cc)) ** 0.5)
    w3 = torch.from_numpy(rng.standard_normal((c3, 2 * Cc, 1, 1)).astype(np.float32) * (2.0 / (2 * Cc)) ** 0.5)
    b1, b2 = (torch.from_numpy(rng.standard_normal(Cc).astype(np.float32) * 0.3) for _ in range(2))
    b3 = torch.from_numpy(rng.standard_normal(c3).astype(np.float32) * 0.3)
    w1p, b1p, _, K1, _ = pack_conv_weight(w1, b1, torch.float16)
    w2p, b2p, _, K2, _ = pack_conv_weight(w2, b2, torch.float16)
    w3p, b3p, _, K3, N3 = pack_conv_weight(w3, b3, torch.float16)
    assert N3 == 64
    xbuf = aligned((B, H, W, ldx), np.float16)
    xbuf[...] = rng.standard_normal(xbuf.shape).astype(np.float16)
    y2buf = aligned((B, H, W, ld2), np.float16)
    y2buf[...] = rng.standard_normal(y2buf.shape).astype(np.float16)
    x, y2 = xbuf[..., ldx - Cc:], y2buf[..., ld2 - Cc:]
    out = aligned((B, H, W, ldo), np.float16, 7)
    bufs = [aligned(t.shape, t.numpy().dtype) for t in (w1p, b1p, w2p, b2p, w3p, b3p)]
    for dst, src in zip(bufs, (w1p, b1p, w2p, b2p, w3p, b3p)):
        dst[...] = src.numpy()
    W1, B1, W2, B2, W3, B3 = bufs
    rc = lib.y5_bottleneck_cv3_fwd(C.c_void_p(xbuf.ctypes.data + (ldx - Cc) * 2), ldx, ptr(W1), ptr(B1), K1, ptr(W2), ptr(B2), K2,
                                   C.c_void_p(y2buf.ctypes.data + (ld2 - Cc) * 2), ld2, ptr(W3), ptr(B3), K3, c3, act3, ptr(out), ldo, B, H, W, Cc, int(add), mb, None)
    assert rc == 0, lib.y5_last_error()
    m = torch.from_numpy(_ref_bneck(np.ascontiguousarray(x), w1, b1, w2, b2, add)).permute(0, 3, 1, 2)
    cat = torch.cat((m, torch.from_numpy(np.ascontiguousarray(y2).astype(np.float32)).permute(0, 3, 1, 2)), 1)
    ref = F.conv2d(cat, w3.half().float(), b3)
    ref = (F.silu(ref) if act3 else ref).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(out[..., :c3].astype(np.float32), ref, rtol=5e-3, atol=5e-3)
    assert np.all(out[..., c3:] == 7)


# c_ = 128 (conv_h3b.h CV3): Bottleneck + cv3 (256 -> <= 256) in one launch; image tails, several tiles per workgroup (the next tile's halo and W1 are issued
# behind GEMM 3), C3 with a channel tail, no activation; both LDS-DMA landing models
CV3_128_CASES = [(1, 6, 7, True, 256, 128, 128, 256, 1, 0), (2, 12, 20, False, 256, 136, 256, 264, 1, 2), (1, 23, 40, True, 248, 128, 128, 256, 0, 2),
                 (3, 9, 10, True, 256, 256, 128, 256, 1, 1)]


def _run_cv3_128(B, H, W, add, c3, ldx, ld2, ldo, act3, mb):
    lib = emu()
    Cc = 128
    rng = np.random.default_rng(H * W + c3 + B)
    w1 = torch.from_numpy(rng.standard_normal((Cc, Cc, 1, 1)).astype(np.float32) * (2.0 / Cc) ** 0.5)
    w2 = torch.from_numpy(rng.standard_normal((Cc, Cc, 3, 3)).astype(np.float32) * (2.0 / (9 * Cc)) ** 0.5)
    w3 = torch.from_numpy(rng.standard_normal((c3, 2 * Cc, 1, 1)).astype(np.float32) * (2.0 / (2 * Cc)) ** 0.5)
    b1, b2 = (torch.from_numpy(rng.standard_normal(Cc).astype(np.float32) * 0.3) for _ in range(2))
    b3 = torch.from_numpy(rng.standard_normal(c3).astype(np.float32) * 0.3)
    w1p, b1p, _, K1, _ = pack_conv_weight(w1, b1, torch.float16)
    w2p, b2p, _, K2, _ = pack_conv_weight(w2, b2, torch.float16)
    w3p, b3p, _, K3, N3 = pack_conv_weight(w3, b3, torch.float16)
    xbuf = aligned((B, H, W, ldx), np.float16)
    xbuf[...] = rng.standard_normal(xbuf.shape).astype(np.float16)
    y2buf = aligned((B, H, W, ld2), np.float16)
    y2buf[...] = rng.standard_normal(y2buf.shape).astype(np.float16)
    x, y2 = xbuf[..., ldx - Cc:], y2buf[..., ld2 - Cc:]
    out = aligned((B, H, W, ldo), np.float16, 7)
    bufs = [aligned(t.shape, t.numpy().dtype) for t in (w1p, b1p, w2p, b2p, w3p, b3p)]
    for dst, src in zip(bufs, (w1p, b1p, w2p, b2p, w3p, b3p)):
        dst[...] = src.numpy()
    W1, B1, W2, B2, W3, B3 = bufs
    rc = lib.y5_bottleneck_cv3_fwd(C.c_void_p(xbuf.ctypes.data + (ldx - Cc) * 2), ldx, ptr(W1), ptr(B1), K1, ptr(W2), ptr(B2), K2,
                                   C.c_void_p(y2buf.ctypes.data + (ld2 - Cc) * 2), ld2, ptr(W3), ptr(B3), K3, c3, act3, ptr(out), ldo, B, H, W, Cc, int(add), mb, None)
    assert rc == 0, lib.y5_last_error()
    m = torch.from_numpy(_ref_bneck(np.ascontiguousarray(x), w1, b1, w2, b2, add)).permute(0, 3, 1, 2)
    cat = torch.cat((m, torch.from_numpy(np.ascontiguousarray(y2).astype(np.float32)).permute(0, 3, 1, 2)), 1)
    ref = F.conv2d(cat, w3.half().float(), b3)
    ref = (F.silu(ref) if act3 else ref).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(out[..., :c3].astype(np.float32), ref, rtol=6e-3, atol=6e-3)
    assert np.all(out[..., c3:] == 7)


@pytest.mark.parametrize("B,H,W,add,c3,ldx,ld2,ldo,act3,mb,async_dma", [c + ("0",) for c in CV3_128_CASES] + [CV3_128_CASES[i] + ("1",) for i in (1, 2)])
def test_fused_bottleneck_cv3_c128_matches_torch(B, H, W, add, c3, ldx, ld2, ldo, act3, mb, async_dma):
    import os
    import subprocess
    import sys

    if async_dma == "1":
        code = f"import tests.test_emu_bneck as t; t._run_cv3_128({B}, {H}, {W}, {add}, {c3}, {ldx}, {ld2}, {ldo}, {act3}, {mb})"
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, Y5_EMU_ASYNC="1"), capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return
    _run_cv3_128(B, H, W, add, c3, ldx, ld2, ldo, act3, mb)


def test_plan_fuses_cv3_into_the_last_bottleneck(monkeypatch):
    """yolov5s 2.C3 (c_ = 32): the plan with Bottleneck + cv3 as one launch (Y5_FUSED_CV3=1) against the two-launch plan on the emulator."""
    from oracle import detgen
    from tests.hipemu.backend import EmuBackend
    from tests.test_emu_model import det_model
    from yolov5_amd.engine import Engine

    m = det_model("yolov5s", 0, True).half()
    x = torch.from_numpy(detgen.uniform((1, 3, 64, 64), 0.0, 1.0, name="img", seed=0)).half()
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("Y5_FUSED_CV3", mode)
        eng = Engine(m, (1, 3, 64, 64), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
        outs[mode] = np.asarray(eng(x)["z"]).astype(np.float32).copy()
        assert any(n.startswith("bneck+cv3:") for n in eng.op_names) == (mode == "1"), eng.op_names
    u, v = outs["0"], outs["1"]
    assert np.abs(u - v).max() <= 4e-3 * max(1.0, np.abs(u).max()), np.abs(u - v).max()


def test_plan_fuses_cv3_into_the_last_128_channel_bottleneck(monkeypatch):
    """yolov5s 6 / 13 / 20.C3 (c_ = 128): the C3 tail (last Bottleneck + cv3) as one conv_h3b.h launch (Y5_EXPERIMENTAL=cv3_128) against the plan with cv3 as
    its own launch, on the emulator (1 x 3 x 64 x 96: 8 x 12, 4 x 6 and 2 x 3 images)."""
    from oracle import detgen
    from tests.hipemu.backend import EmuBackend
    from tests.test_emu_model import det_model
    from yolov5_amd.engine import Engine

    m = det_model("yolov5s", 0, True).half()
    x = torch.from_numpy(detgen.uniform((1, 3, 64, 96), 0.0, 1.0, name="img", seed=0)).half()
    outs = {}
    monkeypatch.setenv("Y5_FUSED_CV3", "0")
    for mode in ("0", "1"):
        monkeypatch.setenv("Y5_EXPERIMENTAL", "cv3_128" if mode == "1" else "")
        monkeypatch.setenv("Y5_FUSED_BNECK128", "force")   # (below the planner's workgroup-count gate at this batch)
        eng = Engine(m, (1, 3, 64, 96), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
        outs[mode] = np.asarray(eng(x)["z"]).astype(np.float32).copy()
        fused = [n for n in eng.op_names if n.startswith("bneck128+cv3:")]
        assert len(fused) == (3 if mode == "1" else 0), eng.op_names
    u, v = outs["0"], outs["1"]
    assert np.abs(u - v).max() <= 4e-3 * max(1.0, np.abs(u).max()), np.abs(u - v).max()


def test_plan_fuses_the_128_channel_bottlenecks(monkeypatch):
    """yolov5s 6.C3 / 13.C3 / 20.C3 (c_ = 128): the plan whose Bottlenecks are conv_h3b.h launches (default) against the two-launch plan
    (Y5_FUSED_BNECK128=0) on the emulator; 13.C3's cv1+cv2 GEMM carries the split store on top of the virtual Upsample + Concat read."""
    from oracle import detgen
    from tests.hipemu.backend import EmuBackend
    from tests.test_emu_model import det_model
    from yolov5_amd.engine import Engine

    m = det_model("yolov5s", 0, True).half()
    x = torch.from_numpy(detgen.uniform((1, 3, 64, 96), 0.0, 1.0, name="img", seed=0)).half()
    outs, names = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("Y5_FUSED_BNECK128", "force" if mode == "1" else mode)   # (force: below the planner's workgroup-count gate at this batch)
        eng = Engine(m, (1, 3, 64, 96), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
        outs[mode] = np.asarray(eng(x)["z"]).astype(np.float32).copy()
        names[mode] = list(eng.op_names)
    n128 = [n for n in names["1"] if n.startswith("bneck128") and n.split(":")[1].split(".")[0] in ("6", "13", "20")]
    assert len(n128) == 5 and not any(n.startswith("bneck128") for n in names["0"]), (names["0"], names["1"])
    assert len(names["1"]) == len(names["0"]) - 5 + 0 or len([n for n in names["1"] if n == "conv:b.cv1"]) == len([n for n in names["0"] if n == "conv:b.cv1"]) - 5
    u, v = outs["0"], outs["1"]
    assert np.abs(u - v).max() <= 4e-3 * max(1.0, np.abs(u).max()), np.abs(u - v).max()
