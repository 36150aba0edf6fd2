"""GPU parity of the training-path kernels through the C-ABI: train-mode BN+SiLU forward/backward, conv weight gradient
(y5_conv2d_wgrad) and conv data gradient (forward kernel on transformed filters with output placement) vs torch-CPU
autograd in fp32 on the same fp16-rounded inputs, at layer sizes of the yolov5s graph."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _st(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("B,H,W,Cc,use_res", [(8, 40, 40, 64, False), (4, 80, 80, 32, True), (16, 20, 20, 256, False)])
def test_bn_silu_fwd_bwd(B, H, W, Cc, use_res, dev):
    from yolov5_amd import _lib

    lib = _lib.lib()
    npix, ld = B * H * W, Cc + 8
    z = torch.from_numpy(detgen.uniform((B, H, W, Cc), -2, 3, name="gz")).half()
    dy = torch.from_numpy(detgen.uniform((B, H, W, Cc), -1, 1, name="gdy")).half()
    res = torch.from_numpy(detgen.uniform((B, H, W, Cc), -1, 1, name="gr")).half() if use_res else None
    gamma = torch.from_numpy(detgen.uniform((Cc,), 0.5, 1.5, name="gg"))
    beta = torch.from_numpy(detgen.uniform((Cc,), -0.5, 0.5, name="gb"))

    def slab(t, fill):
        a = torch.full((npix, ld), fill, dtype=torch.float16, device=dev)
        if t is not None:
            a[:, :Cc] = t.reshape(npix, Cc).to(dev)
        return a

    zd, dyd, rd = slab(z, 3.0), slab(dy, 0.0), slab(res, 0.0) if use_res else None
    yd, dzd = slab(None, -7.0), slab(None, -9.0)
    g, b = gamma.to(dev), beta.to(dev)
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    sm, si, dg, db = (torch.empty(Cc, device=dev) for _ in range(4))
    nws = lib.y5_bn_workspace_bytes(Cc, npix)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    _lib.check(lib.y5_bn_silu_fwd(_p(zd), _lib.Y5_F16, npix, Cc, ld, _p(g), _p(b), 1e-3, 0.03, _p(rm), _p(rv), _p(sm), _p(si), _p(rd), ld,
                                  _p(yd), ld, _p(ws), nws, _st(dev)), lib)
    _lib.check(lib.y5_bn_silu_bwd(_p(dyd), ld, _p(zd), ld, _lib.Y5_F16, npix, Cc, _p(g), _p(b), _p(sm), _p(si), _p(dzd), ld, _p(dg), _p(db),
                                  _p(ws), nws, _st(dev)), lib)
    torch.cuda.synchronize()
    zt = z.float().permute(0, 3, 1, 2).requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    trm, trv = torch.zeros(Cc), torch.ones(Cc)
    out = F.silu(F.batch_norm(zt, trm, trv, gt, bt, True, 0.03, 1e-3))
    if use_res:
        out = out + res.float().permute(0, 3, 1, 2)
    out.backward(dy.float().permute(0, 3, 1, 2))
    tol = dict(rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(yd[:, :Cc].float().cpu().reshape(B, H, W, Cc), out.detach().permute(0, 2, 3, 1), **tol)
    torch.testing.assert_close(dzd[:, :Cc].float().cpu().reshape(B, H, W, Cc), zt.grad.permute(0, 2, 3, 1), **tol)
    torch.testing.assert_close(dg.cpu(), gt.grad, rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(db.cpu(), bt.grad, rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(rm.cpu(), trm, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv.cpu(), trv, rtol=1e-3, atol=1e-5)
    assert torch.all(yd[:, Cc:] == -7.0) and torch.all(dzd[:, Cc:] == -9.0)
    # SyncBatchNorm's split entries with ONE rank (count_total = npix, sums untouched) are the fused entries, bit for bit (train.py:269-271 path;
    # the two-rank exchange itself is tests/test_ddp_gloo.py::test_sync_batchnorm_two_ranks_equal_one_process_full_batch)
    y2, dz2 = slab(None, -7.0), slab(None, -9.0)
    rm2, rv2 = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    sm2, si2, dg2, db2 = (torch.empty(Cc, device=dev) for _ in range(4))
    sums = torch.empty(2 * Cc, dtype=torch.float64, device=dev)
    _lib.check(lib.y5_bn_stats(_p(zd), _lib.Y5_F16, npix, Cc, ld, _p(sums), _p(ws), nws, _st(dev)), lib)
    _lib.check(lib.y5_bn_silu_fwd_from_sums(_p(zd), _lib.Y5_F16, npix, Cc, ld, _p(g), _p(b), 1e-3, 0.03, _p(rm2), _p(rv2), _p(sm2), _p(si2), _p(sums),
                                            npix, _p(rd), ld, _p(y2), ld, _st(dev)), lib)
    _lib.check(lib.y5_bn_bwd_stats(_p(dyd), ld, _p(zd), ld, _lib.Y5_F16, npix, Cc, _p(g), _p(b), _p(sm2), _p(si2), _p(dg2), _p(db2), _p(ws), nws,
                                   _st(dev)), lib)
    _lib.check(lib.y5_bn_silu_bwd_from_sums(_p(dyd), ld, _p(zd), ld, _lib.Y5_F16, npix, Cc, _p(g), _p(b), _p(sm2), _p(si2), _p(dg2), _p(db2), npix,
                                            _p(dz2), ld, _st(dev)), lib)
    torch.cuda.synchronize()
    for a, bb in ((y2, yd), (dz2, dzd), (sm2, sm), (si2, si), (rm2, rm), (rv2, rv), (dg2, dg), (db2, db)):
        assert torch.equal(a, bb)
    np.testing.assert_allclose(sums[:Cc].cpu().numpy() / npix, z.double().reshape(-1, Cc).mean(0).numpy(), rtol=1e-6, atol=1e-7)   # (the per-block partial sums are fp32)


WG = [
    # B, H, W, C1, C2, k, s, p
    (8, 40, 40, 64, 64, (3, 3), (1, 1), (1, 1)),
    (4, 80, 80, 32, 64, (3, 3), (2, 2), (1, 1)),
    (8, 20, 20, 256, 128, (1, 1), (1, 1), (0, 0)),
    (2, 64, 32, 8, 32, (6, 3), (2, 1), (2, 1)),
    (4, 20, 20, 128, 256, (3, 3), (2, 2), (1, 1)),
    (8, 40, 40, 64, 32, (1, 1), (1, 1), (0, 0)),      # pointwise layers: the linear-staging builds (64 x 64, 128 x 128 tiles)
    (8, 20, 20, 256, 136, (1, 1), (1, 1), (0, 0)),
]


# the patch-staged 3x3 family (csrc/wgrad3.h: transpose reads, swizzled tiles, de-interleaved stride-2 patches) on the hardware: (case, cfg, splits, det)
WG3 = [
    ((8, 40, 40, 64, 64, (3, 3), (1, 1), (1, 1)), 3, 0, False),       # 64 x 64 tile (NT 2, CT 2), three segments per row (40 = 16 + 16 + 8)
    ((4, 80, 80, 32, 64, (3, 3), (2, 2), (1, 1)), 3, 37, False),      # stride 2 (33-pixel patch rows, even / odd halves), odd split count
    ((4, 20, 20, 128, 256, (3, 3), (2, 2), (1, 1)), 3, 0, True),      # 128 x 64 tiles x 2 x 2, OW = 10 < 16, deterministic form
    ((2, 36, 52, 128, 136, (3, 3), (1, 1), (1, 1)), 341, 5, True),    # capped tile (NT 4, CT 1), n tail, non-square
    ((2, 33, 47, 40, 72, (3, 3), (2, 2), (1, 1)), 3, 3, False),       # odd sizes, c tail (40 of 64), n tail (72 of 128)
    ((16, 160, 160, 32, 32, (3, 3), (1, 1), (1, 1)), -1, 0, False),   # automatic choice at a P2 shape of the benchmark (many pixels per filter element)
]


@pytest.mark.parametrize("case,cfg,splits,det", WG3)
def test_conv_wgrad_patch_staged(case, cfg, splits, det, dev):
    from yolov5_amd import _lib
    from yolov5_amd.packing import round_up

    B, H, W, C1, C2, k, s, p = case
    lib = _lib.lib()
    OH, OW = (H + 2 - 3) // s[0] + 1, (W + 2 - 3) // s[1] + 1
    x = torch.from_numpy(detgen.uniform((B, C1, H, W), -1, 1, name="g3x")).half()
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="g3dz")).half() * 0.1
    ldx, ldz = C1 + 8, C2 + 8
    xd = torch.full((B, H, W, ldx), 5.0, dtype=torch.float16, device=dev); xd[..., :C1] = x.permute(0, 2, 3, 1).to(dev)
    dzd = torch.full((B, OH, OW, ldz), 5.0, dtype=torch.float16, device=dev); dzd[..., :C2] = dz.permute(0, 2, 3, 1).to(dev)
    K = 9 * C1
    Kpad, Npad = round_up(K, 64), round_up(C2, 32)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=ldx, OH=OH, OW=OW, C2=C2, ldy=ldz, KH=3, KW=3, SH=s[0], SW=s[1],
                      PH=1, PW=1, act=0, Kpad=Kpad, Npad=Npad, cfg=cfg, max_blocks=splits)
    outs = []
    for _ in range(2 if det else 1):
        dw = torch.zeros((Npad, Kpad), dtype=torch.float32, device=dev)
        if det:
            need = lib.y5_conv2d_wgrad_ws_bytes(C.byref(d), ldz)
            ws = torch.full((need // 4,), float("nan"), dtype=torch.float32, device=dev)   # every slab element that is read must have been written
            _lib.check(lib.y5_conv2d_wgrad_det(C.byref(d), _p(xd), _p(dzd), ldz, _p(dw), _p(ws), need, _st(dev)), lib)
        else:
            _lib.check(lib.y5_conv2d_wgrad(C.byref(d), _p(xd), _p(dzd), ldz, _p(dw), _st(dev)), lib)
        torch.cuda.synchronize()
        outs.append(dw)
    if det:
        assert torch.equal(outs[0], outs[1])
    w = torch.zeros((C2, C1, 3, 3), requires_grad=True)
    F.conv2d(x.float(), w, None, s, 1).backward(dz.float())
    ref = w.grad.permute(0, 2, 3, 1).reshape(C2, K)
    scale = float(ref.abs().max())
    torch.testing.assert_close(outs[0][:C2, :K].cpu(), ref, rtol=2e-3, atol=2e-3 * scale)
    assert torch.all(outs[0][C2:] == 0) and torch.all(outs[0][:, K:] == 0)


@pytest.mark.parametrize("B,H,W,C2,splits,det", [(4, 128, 64, 32, 0, False), (2, 66, 41, 32, 7, True), (8, 640, 320, 32, 0, False)])
def test_conv_wgrad_stem_kernel(B, H, W, C2, splits, det, dev):
    """cfg 6 (y5_conv_wgrad_stem_kernel): 0.Conv's weight gradient on the paired-pixel view (k(6,3) s(2,1) p(2,1), 8 channels, ldx = 8) on the MI355X
    vs torch-CPU; the last case is the benchmark's geometry at bs = 8; also == the general kernel (cfg 1) within fp32 association."""
    from yolov5_amd import _lib
    from yolov5_amd.packing import round_up

    lib = _lib.lib()
    OH, OW = (H + 4 - 6) // 2 + 1, W
    x = torch.from_numpy(detgen.uniform((B, 8, H, W), -1, 1, name="gsx")).half()
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="gsdz")).half() * 0.1
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    dzd = dz.permute(0, 2, 3, 1).contiguous().to(dev)
    K, Kpad, Npad = 144, 192, round_up(C2, 32)
    outs = {}
    for cfg in (6, 1):
        d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=8, ldx=8, OH=OH, OW=OW, C2=C2, ldy=C2, KH=6, KW=3, SH=2, SW=1, PH=2, PW=1, act=0,
                          Kpad=Kpad, Npad=Npad, cfg=cfg, max_blocks=splits)
        dw = torch.zeros((Npad, Kpad), dtype=torch.float32, device=dev)
        if det and cfg == 6:
            need = lib.y5_conv2d_wgrad_ws_bytes(C.byref(d), C2)
            ws = torch.full((need // 4,), float("nan"), dtype=torch.float32, device=dev)
            _lib.check(lib.y5_conv2d_wgrad_det(C.byref(d), _p(xd), _p(dzd), C2, _p(dw), _p(ws), need, _st(dev)), lib)
        else:
            _lib.check(lib.y5_conv2d_wgrad(C.byref(d), _p(xd), _p(dzd), C2, _p(dw), _st(dev)), lib)
        torch.cuda.synchronize()
        outs[cfg] = dw.cpu()
    scale = float(outs[1].abs().max())
    torch.testing.assert_close(outs[6], outs[1], rtol=1e-4, atol=1e-4 * scale)
    if B * H * W <= 4 * 128 * 64:
        w = torch.zeros((C2, 8, 6, 3), requires_grad=True)
        F.conv2d(x.float(), w, None, (2, 1), (2, 1)).backward(dz.float())
        ref = w.grad.permute(0, 2, 3, 1).reshape(C2, K)
        torch.testing.assert_close(outs[6][:C2, :K], ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    assert torch.all(outs[6][C2:] == 0) and torch.all(outs[6][:, K:] == 0)


@pytest.mark.parametrize("case", WG)
def test_conv_wgrad(case, dev):
    from yolov5_amd import _lib
    from yolov5_amd.packing import round_up

    B, H, W, C1, C2, k, s, p = case
    lib = _lib.lib()
    OH, OW = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
    x = torch.from_numpy(detgen.uniform((B, C1, H, W), -1, 1, name="gwx")).half()
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="gwdz")).half() * 0.1
    ldx, ldz = C1 + 8, C2 + 8
    xd = torch.full((B, H, W, ldx), 5.0, dtype=torch.float16, device=dev); xd[..., :C1] = x.permute(0, 2, 3, 1).to(dev)
    dzd = torch.full((B, OH, OW, ldz), 5.0, dtype=torch.float16, device=dev); dzd[..., :C2] = dz.permute(0, 2, 3, 1).to(dev)
    K = k[0] * k[1] * C1
    Kpad, Npad = round_up(K, 64), round_up(C2, 32)
    dw = torch.zeros((Npad, Kpad), dtype=torch.float32, device=dev)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=ldx, OH=OH, OW=OW, C2=C2, ldy=ldz, KH=k[0], KW=k[1], SH=s[0], SW=s[1],
                      PH=p[0], PW=p[1], act=0, Kpad=Kpad, Npad=Npad, cfg=-1, max_blocks=0)
    _lib.check(lib.y5_conv2d_wgrad(C.byref(d), _p(xd), _p(dzd), ldz, _p(dw), _st(dev)), lib)
    torch.cuda.synchronize()
    w = torch.zeros((C2, C1, k[0], k[1]), requires_grad=True)
    F.conv2d(x.float(), w, None, s, p).backward(dz.float())
    ref = w.grad.permute(0, 2, 3, 1).reshape(C2, K)
    scale = float(ref.abs().max())
    torch.testing.assert_close(dw[:C2, :K].cpu(), ref, rtol=2e-3, atol=2e-3 * scale)
    assert torch.all(dw[C2:] == 0) and torch.all(dw[:, K:] == 0)


DG = [(8, 40, 40, 64, 64, 3, 1, 1, True), (4, 80, 80, 32, 64, 3, 2, 1, False), (8, 20, 20, 256, 128, 1, 1, 0, False),
      (4, 40, 40, 128, 256, 3, 2, 1, True)]


@pytest.mark.parametrize("case", DG)
def test_conv_dgrad(case, dev):
    from yolov5_amd import _lib
    from yolov5_amd.train_ops import ConvDgrad

    B, H, W, C1, C2, k, s, p, acc = case
    lib = _lib.lib()
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = torch.from_numpy(detgen.uniform((C2, C1, k, k), -0.2, 0.2, name="gdw")).half().float()
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="gddz")).half()
    ldz, ldx = C2 + 8, C1 + 8
    dzd = torch.full((B, OH, OW, ldz), 9.0, dtype=torch.float16, device=dev); dzd[..., :C2] = dz.permute(0, 2, 3, 1).to(dev)
    dxd = torch.zeros((B, H, W, ldx), dtype=torch.float16, device=dev)
    init = torch.from_numpy(detgen.uniform((B, H, W, C1), -1, 1, name="gdx0")).half()
    if acc:
        dxd[..., :C1] = init.to(dev)
    op = ConvDgrad(lib, B, (H, W), (OH, OW), C1, C2, (k, k), (s, s), (p, p))
    op.launch(w.to(dev), dzd.data_ptr(), ldz, dxd.data_ptr(), ldx, acc, _st(dev))
    torch.cuda.synchronize()
    x = torch.zeros((B, C1, H, W), requires_grad=True)
    F.conv2d(x, w, None, s, p).backward(dz.float())
    ref = x.grad.permute(0, 2, 3, 1)
    if acc:
        ref = ref + init.float()
    torch.testing.assert_close(dxd[..., :C1].float().cpu(), ref, rtol=2e-2, atol=3e-2)
    assert torch.all(dxd[..., C1:] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,npix,na,no,ld", [(8, 1024, 3, 85, 256), (3, 64, 3, 85, 256), (4, 400, 3, 85, 264), (2, 6400, 3, 117, 352),
                                              (5, 100, 3, 8, 24)])
def test_head_layout_roundtrip(B, npix, na, no, ld, dev):
    """models/yolo.py:96-98 view/permute and its backward, against the torch permute (bit-exact: pure data movement)."""
    from yolov5_amd import _lib

    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    lg = torch.from_numpy(detgen.uniform((B, npix, ld), -4, 4, name="hl")).half().to(dev)
    raw = torch.full((B, na, npix, no), 9.0, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_nhwc_to_raw(C.c_void_p(lg.data_ptr()), C.c_void_p(raw.data_ptr()), B, npix, na, no, ld, st), lib)
    ref = lg[..., : na * no].view(B, npix, na, no).permute(0, 2, 1, 3).contiguous()
    assert torch.equal(raw, ref)
    draw = torch.from_numpy(detgen.uniform((B, na, npix, no), -1, 1, name="hd")).half().to(dev)
    dlg = torch.full((B, npix, ld), 5.0, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_raw_to_nhwc(C.c_void_p(draw.data_ptr()), C.c_void_p(dlg.data_ptr()), B, npix, na, no, ld, st), lib)
    torch.cuda.synchronize()
    want = torch.zeros_like(dlg)
    want[..., : na * no] = draw.permute(0, 2, 1, 3).reshape(B, npix, na * no)
    assert torch.equal(dlg, want)


@pytest.mark.parametrize("B,H,W,Cc", [(64, 20, 20, 256), (3, 13, 17, 24), (2, 40, 40, 64)])
def test_sppf_pool_bwd_matches_autograd_and_repeats(B, H, W, Cc, dev):
    """y5_sppf_pool_bwd (csrc/train_misc.hip; SPPF's three chained max_pool2d(5, 1, 2), models/common.py:338-340) against torch autograd on the CPU --
    fp16 activations quantised to a coarse grid so that windows hold many equal maxima (the first-in-scan-order tie rule decides) -- and bit-identical
    across repeated launches: the gather form has no atomics (bs 64 x 20 x 20 x 256 is yolov5s' 9.SPPF at 640^2)."""
    from yolov5_amd import _lib

    lib = _lib.lib()
    g = torch.Generator().manual_seed(B + H + Cc)
    x = ((torch.rand((B, Cc, H, W), generator=g) * 8).round() / 4 - 1).half().float().requires_grad_(True)   # 9 distinct values: ties everywhere
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    gs = [(torch.rand((B, Cc, H, W), generator=g) * 2 - 1).half().float() for _ in range(4)]
    (x * gs[0] + y1 * gs[1] + y2 * gs[2] + y3 * gs[3]).sum().backward()
    act = torch.cat([t.detach().permute(0, 2, 3, 1) for t in (x, y1, y2, y3)], -1).half().contiguous().to(dev)
    grad0 = torch.cat([t.permute(0, 2, 3, 1) for t in gs], -1).half().contiguous().to(dev)
    outs = []
    for _ in range(3):
        grad = grad0.clone()
        _lib.check(lib.y5_sppf_pool_bwd(C.c_void_p(act.data_ptr()), C.c_void_p(grad.data_ptr()), B, H, W, Cc, 4 * Cc, 4 * Cc, 5, _lib.stream(dev)), lib)
        torch.cuda.synchronize()
        outs.append(grad[..., :Cc].clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = x.grad.permute(0, 2, 3, 1)
    got = outs[0].float().cpu()
    # sums of up to 1 + 25 + 25^2 ... fp16-rounded terms in fp32, rounded to fp16 once: a few fp16 ulps of the largest sums
    assert float((got - ref).abs().max()) <= 4e-3 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())
    # an fp16 overflow in an incoming slice must stay visible to the loss scaler (the integer grid of the scatter form has no Inf: the workgroup's output is poisoned)
    grad = grad0.clone()
    grad[0, H // 2, W // 2, 2 * Cc + 1] = float("inf")
    _lib.check(lib.y5_sppf_pool_bwd(C.c_void_p(act.data_ptr()), C.c_void_p(grad.data_ptr()), B, H, W, Cc, 4 * Cc, 4 * Cc, 5, _lib.stream(dev)), lib)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(grad[0, ..., :Cc].float()).all())
    assert bool(torch.isfinite(grad[1:, ..., :Cc].float()).all())


@pytest.mark.parametrize("B,H,W,C1,C2,k,s,cfgs", [
    (64, 320, 320, 32, 64, 3, 2, (31, 34, 81)),      # 1.Conv
    (64, 160, 160, 64, 64, 1, 1, (16, 17, 85)),      # 2.C3.cv1+cv2 / cv3
    (64, 160, 160, 32, 32, 1, 1, (14,)),             # 2.C3.m.0.cv1
    (64, 160, 160, 32, 32, 3, 1, (30, 33, 82, 83)),  # 2.C3.m.0.cv2
    (64, 80, 80, 128, 128, 1, 1, (19, 20, 84)),      # 4.C3.cv1+cv2 / cv3
    (64, 80, 80, 64, 64, 3, 1, (32, 78, 79, 80)),    # 4.C3.m.*.cv2
    (64, 80, 80, 128, 64, 1, 1, (18, 21, 86)),       # 17.C3.cv1 ...
])
def test_conv_fwd_stats_at_benchmarked_shapes(B, H, W, C1, C2, k, s, cfgs, dev):
    """the BatchNorm statistics fused into the convolution epilogue (y5_conv2d_fwd_stats; off-switch Y5_DISABLE=bn_fused_stats) had GPU coverage only
    through the wide whole-plan bound.  Here, at the shapes and on every streaming configuration the training plan of yolov5s bs 64 uses it with: z is
    BIT-IDENTICAL to y5_conv2d_fwd's, the per-workgroup partial rows add up to the float64 sums of that z (rtol 1e-5), and y5_bn_silu_fwd_from_partials gives
    the mean / invstd / running statistics of the separate pass (y5_bn_silu_fwd) to 1e-5 and its y to one fp16 ulp (models/common.py:82-88 train mode)."""
    from yolov5_amd import _lib
    from yolov5_amd.packing import pack_conv_weight

    lib = _lib.lib()
    st = _lib.stream(dev)
    vp = lambda t_: C.c_void_p(t_.data_ptr())
    g = torch.Generator().manual_seed(C1 + C2 + k)
    x = torch.randn((B, H, W, C1), generator=g).half().to(dev)
    w = (torch.randn((C2, C1, k, k), generator=g) * (2.0 / (C1 * k * k)) ** 0.5)
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2), torch.float16)
    wp, bp = wp.to(dev), bp.to(dev)
    p = k // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    npix = B * OH * OW
    gamma = (torch.rand(C2, generator=g) + 0.5).to(dev)
    beta = (torch.rand(C2, generator=g) - 0.5).to(dev)
    ran = 0
    for cfg in cfgs:
        d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=C1, OH=OH, OW=OW, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=0,
                          Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=0)
        z0 = torch.full((npix, C2), 5.0, dtype=torch.float16, device=dev)
        if lib.y5_conv2d_fwd(C.byref(d), vp(x), vp(wp), vp(bp), None, vp(z0), None, st) != 0:
            continue                                # configuration not built for this shape
        z1 = torch.full((npix, C2), 5.0, dtype=torch.float16, device=dev)
        part = torch.full((8 * 256 * 2 * C2,), float("nan"), dtype=torch.float32, device=dev)
        rows = C.c_int(0)
        rc = lib.y5_conv2d_fwd_stats(C.byref(d), vp(x), vp(wp), vp(bp), vp(z1), vp(part), part.numel() * 4, C.byref(rows), st)
        assert rc == 0, lib.y5_last_error()
        torch.cuda.synchronize()
        assert torch.equal(z0, z1), cfg
        pr = part[: rows.value * 2 * C2].view(rows.value, 2, C2).double()
        zf = z1.double()
        torch.testing.assert_close(pr[:, 0].sum(0), zf.sum(0), rtol=1e-5, atol=1e-2)
        torch.testing.assert_close(pr[:, 1].sum(0), (zf * zf).sum(0), rtol=1e-5, atol=1e-2)
        outs = []
        for fused in (False, True):
            rm, rv = torch.zeros(C2, device=dev), torch.ones(C2, device=dev)
            sm, si = torch.empty(C2, device=dev), torch.empty(C2, device=dev)
            y = torch.full((npix, C2), -7.0, dtype=torch.float16, device=dev)
            if fused:
                rc = lib.y5_bn_silu_fwd_from_partials(vp(z1), _lib.Y5_F16, npix, C2, C2, vp(gamma), vp(beta), 1e-3, 0.03, vp(rm), vp(rv), vp(sm), vp(si),
                                                      vp(part), rows.value, None, 0, vp(y), C2, st)
            else:
                nws = lib.y5_bn_workspace_bytes(C2, npix)
                ws = torch.empty(nws, dtype=torch.uint8, device=dev)
                rc = lib.y5_bn_silu_fwd(vp(z1), _lib.Y5_F16, npix, C2, C2, vp(gamma), vp(beta), 1e-3, 0.03, vp(rm), vp(rv), vp(sm), vp(si), None, 0,
                                        vp(y), C2, vp(ws), nws, st)
            assert rc == 0, lib.y5_last_error()
            torch.cuda.synchronize()
            outs.append((sm, si, rm, rv, y.float()))
        for u, v in zip(outs[0][:4], outs[1][:4]):
            torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-6)
        dy = (outs[0][4] - outs[1][4]).abs()
        assert float((dy / (outs[0][4].abs() * 2.0 ** -10 + 2.0 ** -14)).max()) <= 1.0, cfg   # one fp16 ulp where the statistics differ in their last bit
        ran += 1
    assert ran >= 1
