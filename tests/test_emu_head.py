"""CPU: fused Detect head (yolov5_amd/csrc/head.hip: y5_detect_head_fwd = 1x1 Detect convolution + decode in one pass of the
streaming pointwise kernel) on the HIP emulator -- bit-identical to y5_conv2d_fwd(act=0) + y5_detect_decode(raw=NULL), and
the engine-level selection (Y5_FUSED_HEAD) on a yolov5s plan in export mode."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import detgen
from tests.hipemu.backend import EmuBackend
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight


def _inputs(B, ny, nx, ldx, seed):
    x = aligned((B, ny, nx, ldx), np.float16, 3.0)
    x[..., :128] = detgen.uniform((B, ny, nx, 128), -1, 1, name="hx", seed=seed).astype(np.float16)
    w = torch.from_numpy(detgen.uniform((255, 128, 1, 1), -0.25, 0.25, name="hw", seed=seed))
    b = torch.from_numpy(detgen.uniform((255,), -2.0, 1.0, name="hb", seed=seed))
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    wp_a = aligned(wp.shape, np.float16); wp_a[...] = wp.numpy()
    bp_a = aligned(bp.shape, np.float32); bp_a[...] = bp.numpy()
    return x, wp_a, bp_a, Kpad, Npad


@pytest.mark.parametrize("B,ny,nx,max_blocks,row_off,extra", [(2, 8, 8, 0, 0, 0), (3, 8, 12, 1, 8, 16), (1, 16, 20, 2, 0, 8), (3, 8, 12, 1 | (87 << 16), 8, 16)])
def test_fused_head_bit_identical_to_conv_plus_decode(B, ny, nx, max_blocks, row_off, extra):
    lib = emu()
    ldx = 136
    x, wp, bp, Kpad, Npad = _inputs(B, ny, nx, ldx, seed=B)
    assert Npad == 256
    npix = ny * nx
    nrows = row_off + 3 * npix + extra
    anchors = (C.c_float * 6)(10.0, 13.0, 16.0, 30.0, 33.0, 23.0)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=ny, W=nx, C1=128, ldx=ldx, OH=ny, OW=nx, C2=256, ldy=256, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0,
                      act=0, Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=(max_blocks >> 16) or 56, max_blocks=max_blocks & 0xffff)  # (87 << 16: the eight-wave kernels)
    # two-call form
    lg = aligned((B, ny, nx, 256), np.float16, -9.0)
    assert lib.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(wp), ptr(bp), None, ptr(lg), None, None) == 0, lib.y5_last_error()
    z_ref = aligned((B, nrows, 85), np.float16, 7.0)
    assert lib.y5_detect_decode(ptr(lg), _lib.Y5_F16, B, ny, nx, 3, 85, 0, 256, 8.0, anchors, ptr(z_ref), _lib.Y5_F16, nrows, row_off, None,
                                None) == 0, lib.y5_last_error()
    # fused
    z = aligned((B, nrows, 85), np.float16, 7.0)
    rc = lib.y5_detect_head_fwd(C.byref(d), ptr(x), ptr(wp), ptr(bp), ny, nx, 8.0, anchors, ptr(z), nrows, row_off, None)
    assert rc == 0, lib.y5_last_error()
    assert np.array_equal(z.view(np.uint16), z_ref.view(np.uint16))
    assert (z[:, :row_off] == 7.0).all() and (z[:, row_off + 3 * npix:] == 7.0).all()  # rows of other levels untouched
    zw = z[:, row_off:row_off + 3 * npix].astype(np.float32)
    assert np.isfinite(zw).all() and zw[..., 4:].max() <= 1.0 and zw[..., 2:4].min() >= 0.0
    # the hint form: same z, plus a plane holding z[..., 4] bit for bit (one more store per anchor block in the counted-vmcnt pipeline)
    z2 = aligned((B, nrows, 85), np.float16, 7.0)
    hint = aligned((B, nrows), np.float16, -3.0)
    rc = lib.y5_detect_head_fwd_hint(C.byref(d), ptr(x), ptr(wp), ptr(bp), ny, nx, 8.0, anchors, ptr(z2), nrows, row_off, ptr(hint), None)
    assert rc == 0, lib.y5_last_error()
    assert np.array_equal(z2.view(np.uint16), z_ref.view(np.uint16))
    sl = slice(row_off, row_off + 3 * npix)
    assert np.array_equal(hint[:, sl].view(np.uint16), z2[:, sl, 4].view(np.uint16))
    assert (hint[:, :row_off] == -3.0).all() and (hint[:, row_off + 3 * npix:] == -3.0).all()


def test_fused_head_rejects_other_shapes():
    lib = emu()
    x, wp, bp, Kpad, Npad = _inputs(1, 5, 5, 128, seed=9)
    anchors = (C.c_float * 6)(1, 2, 3, 4, 5, 6)
    z = aligned((1, 80, 85), np.float16)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=1, H=5, W=5, C1=128, ldx=128, OH=5, OW=5, C2=256, ldy=256, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0,
                      act=0, Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=-1, max_blocks=0)
    assert lib.y5_detect_head_fwd(C.byref(d), ptr(x), ptr(wp), ptr(bp), 5, 5, 8.0, anchors, ptr(z), 80, 0, None) != 0  # 25 pixels: not % 32
    d.H = d.OH = 8; d.W = d.OW = 8; d.act = 1
    assert lib.y5_detect_head_fwd(C.byref(d), ptr(x), ptr(wp), ptr(bp), 8, 8, 8.0, anchors, ptr(z), 192, 0, None) != 0  # activation
    d.act = 0; d.C1 = 64
    assert lib.y5_detect_head_fwd(C.byref(d), ptr(x), ptr(wp), ptr(bp), 8, 8, 8.0, anchors, ptr(z), 192, 0, None) != 0  # 64 input channels
    d.C1 = 128
    assert lib.y5_detect_head_fwd(C.byref(d), ptr(x), ptr(wp), ptr(bp), 8, 8, 8.0, anchors, ptr(z), 196, 4, None) != 0  # z rows not 8-aligned


def test_engine_fused_head_same_outputs(monkeypatch):
    """yolov5s export plan at 64x128 (P3 grid 8x16, 128 channels, 504 z rows): Y5_FUSED_HEAD=1 replaces the level-0 convolution + decode by the fused
    launch (plan indices unchanged: the decode slot becomes a no-op) and z is bit-identical; so does level 1 (round 5: the K-streamed kernel of
    conv_headk.h, 4 x 8 = 32 pixels at 256 channels); level 2 (2 x 4 pixels) keeps the two-op form."""
    from yolov5_amd.engine import Engine
    from yolov5_amd.yolo import DetectionModel

    torch.manual_seed(0)
    m = DetectionModel("yolov5s.yaml").eval().fuse().half()
    x = torch.from_numpy(detgen.uniform((1, 3, 64, 128), 0.0, 1.0, name="img", seed=5)).half()
    monkeypatch.setenv("Y5_FUSED_HEAD", "0")
    a = Engine(m, (1, 3, 64, 128), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
    za = a(x)["z"].copy()
    monkeypatch.setenv("Y5_FUSED_HEAD", "1")
    b = Engine(m, (1, 3, 64, 128), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
    zb = b(x)["z"].copy()
    assert b._fused_heads == {0, 1} and not a._fused_heads
    assert [n for n in b.op_names if "fused" in n or "conv+decode" in n] == ["conv+decode:detect.m0", "decode0(fused)", "conv+decode:detect.m1", "decode1(fused)"]
    assert len(a.op_names) == len(b.op_names)
    assert np.array_equal(za.view(np.uint16), zb.view(np.uint16))
    # raw tensors wanted (eval mode without export): the logits are an output, nothing is fused
    c = Engine(m, (1, 3, 64, 128), torch.float16, "cpu", want_raw=True, backend=EmuBackend())
    assert not c._fused_heads


DEEP_CASES = [(2, 8, 8, 256, 0, 0), (4, 5, 8, 512, 8, 16), (2, 20, 20, 256, 0, 8), (3, 8, 12, 160, 16, 0), (16, 2, 20, 512, 0, 0)]


@pytest.mark.parametrize("B,ny,nx,C1,row_off,extra,async_dma", [c + ("0",) for c in DEEP_CASES] + [DEEP_CASES[i] + ("1",) for i in (1, 2)])
def test_fused_head_deep_levels_bit_identical(B, ny, nx, C1, row_off, extra, async_dma):
    """y5_detect_head_fwd for C1 > 128 (csrc/conv_headk.h: K streamed through an LDS ring; P4 / P5 of yolov5s): bit-identical to y5_conv2d_fwd(act = 0) +
    y5_detect_decode; 5 x 8 = 40 and 20 x 20 = 400 pixels per image put image boundaries INSIDE 32-pixel wave tiles (two store segments)."""
    import os
    import subprocess
    import sys

    if async_dma == "1":
        code = f"import tests.test_emu_head as t; t._run_deep({B}, {ny}, {nx}, {C1}, {row_off}, {extra})"
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, Y5_EMU_ASYNC="1"), capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return
    _run_deep(B, ny, nx, C1, row_off, extra)


def _run_deep(B, ny, nx, C1, row_off, extra):
    lib = emu()
    ldx = C1 + 8
    x = aligned((B, ny, nx, ldx), np.float16, 3.0)
    x[..., :C1] = detgen.uniform((B, ny, nx, C1), -1, 1, name="hx", seed=B + C1).astype(np.float16)
    w = torch.from_numpy(detgen.uniform((255, C1, 1, 1), -0.2, 0.2, name="hw", seed=C1))
    b = torch.from_numpy(detgen.uniform((255,), -2.0, 1.0, name="hb", seed=C1))
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    wp_a = aligned(wp.shape, np.float16); wp_a[...] = wp.numpy()
    bp_a = aligned(bp.shape, np.float32); bp_a[...] = bp.numpy()
    npix = ny * nx
    nrows = row_off + 3 * npix + extra
    anchors = (C.c_float * 6)(30.0, 61.0, 62.0, 45.0, 59.0, 119.0)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=ny, W=nx, C1=C1, ldx=ldx, OH=ny, OW=nx, C2=256, ldy=256, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0,
                      act=0, Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=2, max_blocks=0)
    lg = aligned((B, ny, nx, 256), np.float16, -9.0)
    assert lib.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(wp_a), ptr(bp_a), None, ptr(lg), None, None) == 0, lib.y5_last_error()
    z_ref = aligned((B, nrows, 85), np.float16, 7.0)
    assert lib.y5_detect_decode(ptr(lg), _lib.Y5_F16, B, ny, nx, 3, 85, 0, 256, 16.0, anchors, ptr(z_ref), _lib.Y5_F16, nrows, row_off, None,
                                None) == 0, lib.y5_last_error()
    for with_hint in (False, True):
        z = aligned((B, nrows, 85), np.float16, 7.0)
        hint = aligned((B, nrows), np.float16, -3.0)
        rc = lib.y5_detect_head_fwd_hint(C.byref(d), ptr(x), ptr(wp_a), ptr(bp_a), ny, nx, 16.0, anchors, ptr(z), nrows, row_off, ptr(hint) if with_hint else None, None)
        assert rc == 0, lib.y5_last_error()
        # (the K loop adds the 32-channel chunks in the same order as the implicit GEMM of configuration 2: same fp32 sums, same fp16 logits)
        dz = np.abs(z.astype(np.float32) - z_ref.astype(np.float32))
        assert dz.max() <= 2e-3 * max(1.0, float(np.abs(z_ref.astype(np.float32)).max())), dz.max()
        assert (z[:, :row_off] == 7.0).all() and (z[:, row_off + 3 * npix:] == 7.0).all()
        if with_hint:
            sl = slice(row_off, row_off + 3 * npix)
            assert np.array_equal(hint[:, sl].view(np.uint16), z[:, sl, 4].view(np.uint16))
            assert (hint[:, :row_off] == -3.0).all() and (hint[:, row_off + 3 * npix:] == -3.0).all()
