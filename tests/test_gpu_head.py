"""GPU: fused Detect head (HIP y5_detect_head_fwd) bit-identical to y5_conv2d_fwd(act=0) + y5_detect_decode(raw=NULL) at the P3
shape of 640x640 inputs, and through the engine (Y5_FUSED_HEAD=0 / 1) on yolov5s in export mode."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import detgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,ny,nx,max_blocks,row_off,extra", [(4, 80, 80, 0, 0, 6000), (2, 8, 12, 1, 8, 16), (3, 40, 40, 0, 0, 0), (4, 80, 80, 87 << 16, 0, 6000)])
def test_fused_head_bit_identical(B, ny, nx, max_blocks, row_off, extra, dev):
    from yolov5_amd import _lib
    from yolov5_amd.packing import pack_conv_weight

    lib = _lib.lib()
    ldx = 136
    x = torch.full((B, ny, nx, ldx), 3.0, dtype=torch.float16)
    x[..., :128] = torch.from_numpy(detgen.uniform((B, ny, nx, 128), -1, 1, name="hx", seed=B)).half()
    w = torch.from_numpy(detgen.uniform((255, 128, 1, 1), -0.25, 0.25, name="hw", seed=B))
    b = torch.from_numpy(detgen.uniform((255,), -2.0, 1.0, name="hb", seed=B))
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    x, wp, bp = x.to(dev), wp.to(dev), bp.to(dev)
    npix = ny * nx
    nrows = row_off + 3 * npix + extra
    anchors = (C.c_float * 6)(10.0, 13.0, 16.0, 30.0, 33.0, 23.0)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=ny, W=nx, C1=128, ldx=ldx, OH=ny, OW=nx, C2=256, ldy=256, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0,
                      act=0, Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=(max_blocks >> 16) or 56, max_blocks=max_blocks & 0xffff)  # (87 << 16: the eight-wave kernels)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    lg = torch.full((B, ny, nx, 256), -9.0, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_conv2d_fwd(C.byref(d), p(x), p(wp), p(bp), None, p(lg), None, st), lib)
    z_ref = torch.full((B, nrows, 85), 7.0, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_detect_decode(p(lg), _lib.Y5_F16, B, ny, nx, 3, 85, 0, 256, 8.0, anchors, p(z_ref), _lib.Y5_F16, nrows, row_off, None, st), lib)
    z = torch.full((B, nrows, 85), 7.0, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_detect_head_fwd(C.byref(d), p(x), p(wp), p(bp), ny, nx, 8.0, anchors, p(z), nrows, row_off, st), lib)
    torch.cuda.synchronize()
    assert torch.equal(z.view(torch.int16), z_ref.view(torch.int16))
    assert bool((z[:, :row_off] == 7.0).all()) and bool((z[:, row_off + 3 * npix:] == 7.0).all())


def test_engine_fused_head_same_outputs(dev, monkeypatch):
    from yolov5_amd.engine import Engine
    from yolov5_amd.yolo import DetectionModel

    torch.manual_seed(0)
    m = DetectionModel("yolov5s.yaml").eval().fuse().half().to(dev)
    x = torch.from_numpy(detgen.uniform((2, 3, 640, 640), 0.0, 1.0, name="img", seed=5)).half().to(dev)
    with torch.no_grad():
        monkeypatch.setenv("Y5_FUSED_HEAD", "0")
        a = Engine(m, (2, 3, 640, 640), torch.float16, dev, want_raw=False)
        za = a(x)["z"].clone()
        monkeypatch.setenv("Y5_FUSED_HEAD", "1")
        b = Engine(m, (2, 3, 640, 640), torch.float16, dev, want_raw=False)
        zb = b(x)["z"].clone()
        zb2 = b(x)["z"].clone()  # hipGraph replay
    torch.cuda.synchronize()
    assert b._fused_heads == {0} and not a._fused_heads
    k = a.op_names.index("conv:detect.m0")
    if a.conv_cfgs[sum(1 for n in a.op_names[:k] if n.startswith("conv"))] == 56:
        assert torch.equal(za.view(torch.int16), zb.view(torch.int16))  # same tile configuration -> same logits bit for bit
    else:
        torch.testing.assert_close(zb.float(), za.float(), rtol=2e-2, atol=0.5)
    assert torch.equal(zb, zb2)
