"""CPU (emulator): the stream-K implicit-GEMM configurations (conv_igemm.h SK, ids 57..60) -- every workgroup owns an equal,
contiguous range of (tile, K-chunk) units; tiles split between workgroups are combined through the registered workspace -- against
the plain configuration of the same tile shape, for grids that cut tiles in two and in three, more workgroups than tiles, 1x1 and
3x3 / strided layers, residual epilogue, and repeated launches (the flags must clean themselves)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight


@pytest.fixture
def lib():
    L = emu()
    n = int(L.y5_conv_sk_workspace_bytes())
    ws = aligned((n,), np.uint8, 0xAB)                      # garbage: only the flag page is cleared by the library
    assert L.y5_conv_set_sk_workspace(ptr(ws), n, None) == 0, L.y5_last_error()
    yield L
    L.y5_conv_set_sk_workspace(None, 0, None)


def _run(L, cfg, mb, x, wp, bp, res, B, H, W, C1, C2, k, s, Kpad, Npad):
    p = k // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = aligned((B, OH, OW, C2), np.float16, 5)
    if res is not None:
        y[...] = res
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=C1, OH=OH, OW=OW, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                      Kpad=Kpad, Npad=Npad, ldr=C2 if res is not None else 0, ld2=0, cfg=cfg, max_blocks=mb)
    rc = L.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(wp), ptr(bp), ptr(y) if res is not None else None, ptr(y), None, None)
    assert rc == 0, L.y5_last_error()
    return y


@pytest.mark.parametrize("sk,base,mb", [(57, 8, 3), (57, 8, 5), (57, 8, 7), (57, 8, 13), (59, 12, 3), (60, 9, 5), (58, 39, 3)])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 128, 3, 1, False), (1, 24, 24, 128, 256, 1, 1, False), (2, 16, 16, 64, 128, 3, 2, False),
                                   (1, 16, 16, 128, 128, 3, 1, True)])
def test_streamk_equals_plain_tiles(lib, sk, base, mb, shape):
    B, H, W, C1, C2, k, s, with_res = shape
    rng = np.random.default_rng(B * 1000 + C1 + k + s)
    w = torch.from_numpy(rng.standard_normal((C2, C1, k, k)).astype(np.float32) * (2.0 / (C1 * k * k)) ** 0.5)
    b = torch.from_numpy(rng.standard_normal(C2).astype(np.float32) * 0.2)
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    x = aligned((B, H, W, C1), np.float16)
    x[...] = rng.standard_normal(x.shape).astype(np.float16)
    Wp, Bp = aligned(wp.shape, np.float16), aligned(bp.shape, np.float32)
    Wp[...] = wp.numpy(); Bp[...] = bp.numpy()
    p = k // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rng.standard_normal((B, OH, OW, C2)).astype(np.float16) if with_res else None
    ref = _run(lib, base, 0, x, Wp, Bp, res, B, H, W, C1, C2, k, s, Kpad, Npad)
    for rep in range(2):                                   # twice: the reader resets every flag it consumed
        got = _run(lib, sk, mb, x, Wp, Bp, res, B, H, W, C1, C2, k, s, Kpad, Npad)
        # same products, another summation order across the K split (fp32), one fp16 rounding at the end
        np.testing.assert_allclose(got.astype(np.float32), ref.astype(np.float32), rtol=2e-3, atol=2e-3)


def test_streamk_needs_workspace():
    L = emu()
    L.y5_conv_set_sk_workspace(None, 0, None)
    x = aligned((1, 8, 8, 64), np.float16)
    w = aligned((128, 576), np.float16)
    b = aligned((128,), np.float32)
    y = aligned((1, 8, 8, 128), np.float16)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=1, H=8, W=8, C1=64, ldx=64, OH=8, OW=8, C2=128, ldy=128, KH=3, KW=3, SH=1, SW=1, PH=1, PW=1, act=1,
                      Kpad=576, Npad=128, ldr=0, ld2=0, cfg=57, max_blocks=0)
    assert L.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(w), ptr(b), None, ptr(y), None, None) != 0
    assert b"workspace" in L.y5_last_error()
