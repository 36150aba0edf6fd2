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
    assert b._fused_heads == {0, 1, 2} and not a._fused_heads   # (round 5: P4 / P5 through the K-streamed kernel of conv_headk.h)
    k = a.op_names.index("conv:detect.m0")
    n3 = 3 * 80 * 80   # rows of level 0
    if a.conv_cfgs[sum(1 for n in a.op_names[:k] if n.startswith("conv"))] == 56:
        assert torch.equal(za[:, :n3].view(torch.int16), zb[:, :n3].view(torch.int16))  # same tile configuration -> same logits bit for bit
    torch.testing.assert_close(zb.float(), za.float(), rtol=2e-2, atol=0.5)   # (other tile shapes add the K chunks in another order: fp16 ulps of the logits)
    assert torch.equal(zb, zb2)


@pytest.mark.parametrize("B,ny,nx,C1,row_off,extra", [(64, 40, 40, 256, 19200, 400), (64, 20, 20, 512, 24000, 0), (4, 20, 20, 320, 0, 8), (16, 80, 80, 192, 0, 0)])
def test_fused_head_deep_levels(B, ny, nx, C1, row_off, extra, dev):
    """y5_detect_head_fwd_hint for C1 > 128 (csrc/conv_headk.h: K streamed; models/yolo.py:91-115 at P4 / P5) against y5_conv2d_fwd(act = 0) + y5_detect_decode:
    the benchmarked shapes (64 x 40 x 40 x 256 and 64 x 20 x 20 x 512 inside the 25 200-row z: image boundaries inside wave tiles at P5) and odd ones."""
    from yolov5_amd import _lib
    from yolov5_amd.packing import pack_conv_weight

    lib = _lib.lib()
    g = torch.Generator().manual_seed(B + C1)
    x = (torch.rand((B, ny, nx, C1), generator=g) * 2 - 1).half().to(dev)
    w = (torch.rand((255, C1, 1, 1), generator=g) - 0.5) * 0.3
    b = torch.rand(255, generator=g) * 3 - 2
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    wp, bp = wp.to(dev), bp.to(dev)
    npix = ny * nx
    nrows = row_off + 3 * npix + extra
    anchors = (C.c_float * 6)(30.0, 61.0, 62.0, 45.0, 59.0, 119.0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    z = torch.full((B, nrows, 85), 7.0, dtype=torch.float16, device=dev)
    hint = torch.full((B, nrows), -3.0, dtype=torch.float16, device=dev)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=ny, W=nx, C1=C1, ldx=C1, OH=ny, OW=nx, C2=256, ldy=256, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0,
                      act=0, Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=2, max_blocks=0)
    _lib.check(lib.y5_detect_head_fwd_hint(C.byref(d), p(x), p(wp), p(bp), ny, nx, 16.0, anchors, p(z), nrows, row_off, p(hint), st), lib)
    lg = torch.full((B, ny, nx, 256), -9.0, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_conv2d_fwd(C.byref(d), p(x), p(wp), p(bp), None, p(lg), None, st), lib)
    z_ref = torch.full((B, nrows, 85), 7.0, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_detect_decode(p(lg), _lib.Y5_F16, B, ny, nx, 3, 85, 0, 256, 16.0, anchors, p(z_ref), _lib.Y5_F16, nrows, row_off, None, st), lib)
    torch.cuda.synchronize()
    zf, rf = z.float(), z_ref.float()
    assert float((zf - rf).abs().max()) <= 2e-3 * max(1.0, float(rf.abs().max()))
    assert bool((z[:, :row_off] == 7.0).all()) and bool((z[:, row_off + 3 * npix:] == 7.0).all())
    sl = slice(row_off, row_off + 3 * npix)
    assert torch.equal(hint[:, sl].view(torch.int16), z[:, sl, 4].contiguous().view(torch.int16))
    assert bool((hint[:, :row_off] == -3.0).all()) and bool((hint[:, row_off + 3 * npix:] == -3.0).all())
    # against torch fp32 on the fp16 operands: logits -> sigmoid -> decode (models/yolo.py:102-111), first image
    lgt = torch.nn.functional.conv2d(x[:1].float().permute(0, 3, 1, 2), w.half().float().to(dev), b.to(dev)).half().float()   # (1, 255, ny, nx)
    y = lgt.view(1, 3, 85, ny, nx).permute(0, 1, 3, 4, 2).sigmoid()
    gy, gx = torch.meshgrid(torch.arange(ny, device=dev).float(), torch.arange(nx, device=dev).float(), indexing="ij")
    grid = torch.stack((gx, gy), -1) - 0.5
    anc = torch.tensor(list(anchors), device=dev).view(3, 1, 1, 2)
    xy = (y[..., :2] * 2 + grid) * 16.0
    wh = (y[..., 2:4] * 2) ** 2 * anc
    ref = torch.cat((xy, wh, y[..., 4:]), -1).view(1, 3 * npix, 85)
    err = (zf[:1, sl] - ref).abs()
    # z is fp16: xy (up to 16 * nx pixels) carries half an fp16 ulp of its magnitude, wh a relative 2^-11, scores an absolute 2^-11 (+ the fast sigmoid)
    assert float((err[..., :2] / ref[..., :2].abs().clamp(min=64.0)).max()) <= 1.5e-3 and float((err[..., 2:4] / ref[..., 2:4].clamp(min=1.0)).max()) <= 3e-3
    assert float(err[..., 4:].max()) <= 2e-3
