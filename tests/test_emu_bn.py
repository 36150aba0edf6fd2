"""CPU: train-mode BatchNorm + SiLU kernels (yolov5_amd/csrc/bn_kernels.h) on the HIP emulator vs torch autograd
(the reference's Conv.forward = act(bn(conv(x))), models/common.py:82-88, BN eps 1e-3 / momentum 0.03)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib


def run_bn(z_nhwc, gamma, beta, dy, res, dtype, ld_extra=8):
    lib = emu()
    B, H, W, Cc = z_nhwc.shape
    npix = B * H * W
    ld = Cc + ld_extra
    dt = _lib.Y5_F16 if dtype == np.float16 else _lib.Y5_F32
    z = aligned((npix, ld), dtype, 3.0); z[:, :Cc] = z_nhwc.reshape(npix, Cc)
    y = aligned((npix, ld), dtype, -7.0)
    r = None
    if res is not None:
        r = aligned((npix, ld), dtype, 0.0); r[:, :Cc] = res.reshape(npix, Cc)
    g = aligned((Cc,), np.float32); g[...] = gamma
    b = aligned((Cc,), np.float32); b[...] = beta
    rm = aligned((Cc,), np.float32, 0.0); rv = aligned((Cc,), np.float32, 1.0)
    sm = aligned((Cc,), np.float32); si = aligned((Cc,), np.float32)
    nws = lib.y5_bn_workspace_bytes(Cc, npix)
    ws = aligned((nws,), np.uint8)
    rc = lib.y5_bn_silu_fwd(ptr(z), dt, npix, Cc, ld, ptr(g), ptr(b), 1e-3, 0.03, ptr(rm), ptr(rv), ptr(sm), ptr(si), ptr(r), ld,
                            ptr(y), ld, ptr(ws), nws, None)
    assert rc == 0, lib.y5_last_error()
    d = aligned((npix, ld), dtype, 0.0); d[:, :Cc] = dy.reshape(npix, Cc)
    dz = aligned((npix, ld), dtype, -9.0)
    dg = aligned((Cc,), np.float32); db = aligned((Cc,), np.float32)
    rc = lib.y5_bn_silu_bwd(ptr(d), ld, ptr(z), ld, dt, npix, Cc, ptr(g), ptr(b), ptr(sm), ptr(si), ptr(dz), ld, ptr(dg), ptr(db),
                            ptr(ws), nws, None)
    assert rc == 0, lib.y5_last_error()
    bs = aligned((Cc,), np.float32)
    rc = lib.y5_channel_sum(ptr(d), dt, npix, Cc, ld, ptr(bs), ptr(ws), nws, None)
    assert rc == 0, lib.y5_last_error()
    assert np.all(y[:, Cc:] == -7.0) and np.all(dz[:, Cc:] == -9.0)
    return y[:, :Cc].reshape(B, H, W, Cc), dz[:, :Cc].reshape(B, H, W, Cc), dg, db, rm, rv, sm, si, bs


@pytest.mark.parametrize("B,H,W,Cc,dtype,use_res", [(2, 5, 7, 16, np.float32, False), (3, 8, 8, 32, np.float32, True),
                                                    (2, 9, 6, 48, np.float16, False), (4, 16, 16, 64, np.float16, True)])
def test_emu_bn_silu_fwd_bwd_vs_torch(B, H, W, Cc, dtype, use_res):
    z = detgen.uniform((B, H, W, Cc), -2, 3, name="bnz").astype(dtype)
    dy = detgen.uniform((B, H, W, Cc), -1, 1, name="bndy").astype(dtype)
    res = detgen.uniform((B, H, W, Cc), -1, 1, name="bnr").astype(dtype) if use_res else None
    gamma = detgen.uniform((Cc,), 0.5, 1.5, name="bng")
    beta = detgen.uniform((Cc,), -0.5, 0.5, name="bnb")
    y, dz, dg, db, rm, rv, sm, si, bs = run_bn(z, gamma, beta, dy, res, dtype)
    zt = torch.from_numpy(z.astype(np.float32)).permute(0, 3, 1, 2).requires_grad_(True)
    gt = torch.from_numpy(gamma).requires_grad_(True)
    bt = torch.from_numpy(beta).requires_grad_(True)
    trm, trv = torch.zeros(Cc), torch.ones(Cc)
    out = F.silu(F.batch_norm(zt, trm, trv, gt, bt, True, 0.03, 1e-3))
    if use_res:
        out = out + torch.from_numpy(res.astype(np.float32)).permute(0, 3, 1, 2)
    out.backward(torch.from_numpy(dy.astype(np.float32)).permute(0, 3, 1, 2))
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == np.float32 else dict(rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(y.astype(np.float32), out.detach().permute(0, 2, 3, 1).numpy(), **tol)
    np.testing.assert_allclose(dz.astype(np.float32), zt.grad.permute(0, 2, 3, 1).numpy(), **tol)
    np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rm, trm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv, trv.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bs, dy.astype(np.float32).reshape(-1, Cc).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_emu_sync_bn_split_entries_two_shards_equal_full_batch(dtype):
    """SyncBatchNorm at the C-ABI (train.py:269-271): two 'ranks' with half of the pixels each -- y5_bn_stats per shard, the host adds the sums
    (what the all-reduce does), y5_bn_silu_fwd_from_sums / y5_bn_bwd_stats + y5_bn_silu_bwd_from_sums with the global count -- against the fused
    entries on the whole batch: outputs, saved and running statistics, dz; dgamma / dbeta add up."""
    lib = emu()
    B, H, W, Cc = 4, 6, 5, 16
    npix, half, ld = B * H * W, B * H * W // 2, Cc + 8
    dt = _lib.Y5_F16 if dtype == np.float16 else _lib.Y5_F32
    zn = detgen.uniform((npix, Cc), -2, 3, name="sz").astype(dtype)
    dyn = detgen.uniform((npix, Cc), -1, 1, name="sdy").astype(dtype)
    z = aligned((npix, ld), dtype, 3.0); z[:, :Cc] = zn
    d = aligned((npix, ld), dtype, 0.0); d[:, :Cc] = dyn
    g = aligned((Cc,), np.float32); g[...] = detgen.uniform((Cc,), 0.5, 1.5, name="sg")
    b = aligned((Cc,), np.float32); b[...] = detgen.uniform((Cc,), -0.5, 0.5, name="sb")
    nws = lib.y5_bn_workspace_bytes(Cc, npix)
    ws = aligned((nws,), np.uint8)

    def stats():
        return (aligned((Cc,), np.float32, 0.0), aligned((Cc,), np.float32, 1.0), aligned((Cc,), np.float32), aligned((Cc,), np.float32))

    rm, rv, sm, si = stats()
    y = aligned((npix, ld), dtype, -7.0); dz = aligned((npix, ld), dtype, -9.0)
    dg = aligned((Cc,), np.float32); db = aligned((Cc,), np.float32)
    assert lib.y5_bn_silu_fwd(ptr(z), dt, npix, Cc, ld, ptr(g), ptr(b), 1e-3, 0.03, ptr(rm), ptr(rv), ptr(sm), ptr(si), None, 0, ptr(y), ld, ptr(ws), nws, None) == 0
    assert lib.y5_bn_silu_bwd(ptr(d), ld, ptr(z), ld, dt, npix, Cc, ptr(g), ptr(b), ptr(sm), ptr(si), ptr(dz), ld, ptr(dg), ptr(db), ptr(ws), nws, None) == 0
    # two shards
    es = z.itemsize
    shard = [(z.ctypes.data + k * half * ld * es, d.ctypes.data + k * half * ld * es) for k in (0, 1)]
    sums = [aligned((2 * Cc,), np.float64) for _ in (0, 1)]
    for k, (zp, _) in enumerate(shard):
        assert lib.y5_bn_stats(C.c_void_p(zp), dt, half, Cc, ld, ptr(sums[k]), ptr(ws), nws, None) == 0
    tot = aligned((2 * Cc,), np.float64); tot[...] = sums[0] + sums[1]                 # the all-reduce
    y2 = aligned((npix, ld), dtype, -7.0); dz2 = aligned((npix, ld), dtype, -9.0)
    per = []
    for k, (zp, dp) in enumerate(shard):
        rm2, rv2, sm2, si2 = stats()
        yo_ = y2.ctypes.data + k * half * ld * es
        assert lib.y5_bn_silu_fwd_from_sums(C.c_void_p(zp), dt, half, Cc, ld, ptr(g), ptr(b), 1e-3, 0.03, ptr(rm2), ptr(rv2), ptr(sm2), ptr(si2), ptr(tot),
                                            npix, None, 0, C.c_void_p(yo_), ld, None) == 0, lib.y5_last_error()
        dgk = aligned((Cc,), np.float32); dbk = aligned((Cc,), np.float32)
        assert lib.y5_bn_bwd_stats(C.c_void_p(dp), ld, C.c_void_p(zp), ld, dt, half, Cc, ptr(g), ptr(b), ptr(sm2), ptr(si2), ptr(dgk), ptr(dbk), ptr(ws), nws, None) == 0
        per.append((rm2, rv2, sm2, si2, dgk, dbk))
    gsum = aligned((2 * Cc,), np.float32)
    gsum[:Cc] = per[0][4] + per[1][4]; gsum[Cc:] = per[0][5] + per[1][5]                # the all-reduce of the backward
    for k, (zp, dp) in enumerate(shard):
        dzo = dz2.ctypes.data + k * half * ld * es
        assert lib.y5_bn_silu_bwd_from_sums(C.c_void_p(dp), ld, C.c_void_p(zp), ld, dt, half, Cc, ptr(g), ptr(b), ptr(per[k][2]), ptr(per[k][3]), ptr(gsum),
                                            C.c_void_p(gsum.ctypes.data + 4 * Cc), npix, C.c_void_p(dzo), ld, None) == 0, lib.y5_last_error()
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == np.float16 else dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y2[:, :Cc].astype(np.float32), y[:, :Cc].astype(np.float32), **tol)
    np.testing.assert_allclose(dz2[:, :Cc].astype(np.float32), dz[:, :Cc].astype(np.float32), **tol)
    for k in (0, 1):
        for a, ref in zip(per[k][:4], (rm, rv, sm, si)):
            np.testing.assert_allclose(a, ref, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gsum[:Cc], dg, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gsum[Cc:], db, rtol=1e-4, atol=1e-5)
    assert np.all(y2[:, Cc:] == -7.0) and np.all(dz2[:, Cc:] == -9.0)


@pytest.mark.parametrize("B,H,W,C1,C2,k,cfg,mb", [(7, 8, 16, 32, 32, 1, 14, 1), (5, 8, 16, 128, 64, 1, 18, 2), (5, 8, 16, 128, 128, 1, 84, 1),
                                                  (3, 8, 16, 128, 248, 1, 87, 2), (2, 8, 16, 32, 32, 3, 30, 1), (3, 8, 16, 64, 64, 3, 80, 1),
                                                  (1, 12, 24, 64, 56, 3, 79, 2)])
def test_conv_fwd_stats_equals_the_separate_statistics_pass(B, H, W, C1, C2, k, cfg, mb):
    """y5_conv2d_fwd_stats (streaming pointwise / 3x3 kernels, act = 0): z identical to y5_conv2d_fwd's, and y5_bn_silu_fwd_from_partials on its
    per-workgroup partial rows gives the mean / invstd / running statistics / y of y5_bn_silu_fwd on that z (models/common.py:82-88 train mode)."""
    from tests.test_emu_conv import run_conv  # noqa: F401  (same packing helpers)
    from yolov5_amd.packing import pack_conv_weight

    lib = emu()
    x = torch.from_numpy(detgen.uniform((B, C1, H, W), -1, 1, name="sx")).half().float()
    w = torch.from_numpy(detgen.uniform((C2, C1, k, k), -0.3, 0.3, name="sw")).half().float()
    b = torch.zeros(C2)
    p = k // 2
    xa = aligned((B, H, W, C1), np.float16); xa[...] = x.permute(0, 2, 3, 1).numpy().astype(np.float16)
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    wa = aligned(wp.shape, np.float16); wa[...] = wp.numpy()
    ba = aligned(bp.shape, np.float32); ba[...] = bp.numpy()
    npix = B * H * W
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=C1, OH=H, OW=W, C2=C2, ldy=C2, KH=k, KW=k, SH=1, SW=1, PH=p, PW=p, act=0,
                      Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=mb)
    z0 = aligned((npix, C2), np.float16, 5.0)
    assert lib.y5_conv2d_fwd(C.byref(d), ptr(xa), ptr(wa), ptr(ba), None, ptr(z0), None, None) == 0, lib.y5_last_error()
    z1 = aligned((npix, C2), np.float16, 5.0)
    part = aligned((64 * 2 * C2,), np.float32, np.nan)
    rows = C.c_int(0)
    rc = lib.y5_conv2d_fwd_stats(C.byref(d), ptr(xa), ptr(wa), ptr(ba), ptr(z1), ptr(part), part.nbytes, C.byref(rows), None)
    assert rc == 0, lib.y5_last_error()
    assert np.array_equal(z0, z1) and 1 <= rows.value <= 64
    pr = part[:rows.value * 2 * C2].reshape(rows.value, 2, C2).astype(np.float64)
    zf = z1.astype(np.float64)
    np.testing.assert_allclose(pr[:, 0].sum(0), zf.sum(0), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(pr[:, 1].sum(0), (zf * zf).sum(0), rtol=1e-5, atol=1e-3)
    g = aligned((C2,), np.float32); g[...] = detgen.uniform((C2,), 0.5, 1.5, name="sg")
    be = aligned((C2,), np.float32); be[...] = detgen.uniform((C2,), -0.5, 0.5, name="sb")
    outs = []
    for fused in (False, True):
        rm = aligned((C2,), np.float32, 0.0); rv = aligned((C2,), np.float32, 1.0)
        sm = aligned((C2,), np.float32); si = aligned((C2,), np.float32)
        y = aligned((npix, C2), np.float16, -7.0)
        if fused:
            rc = lib.y5_bn_silu_fwd_from_partials(ptr(z1), _lib.Y5_F16, npix, C2, C2, ptr(g), ptr(be), 1e-3, 0.03, ptr(rm), ptr(rv), ptr(sm), ptr(si),
                                                  ptr(part), rows.value, None, 0, ptr(y), C2, None)
        else:
            nws = lib.y5_bn_workspace_bytes(C2, npix)
            ws = aligned((nws,), np.uint8)
            rc = lib.y5_bn_silu_fwd(ptr(z1), _lib.Y5_F16, npix, C2, C2, ptr(g), ptr(be), 1e-3, 0.03, ptr(rm), ptr(rv), ptr(sm), ptr(si), None, 0,
                                    ptr(y), C2, ptr(ws), nws, None)
        assert rc == 0, lib.y5_last_error()
        outs.append((sm.copy(), si.copy(), rm.copy(), rv.copy(), y.astype(np.float32)))
    for u, v in zip(outs[0][:4], outs[1][:4]):
        np.testing.assert_allclose(u, v, rtol=2e-6, atol=1e-7)
    assert np.abs(outs[0][4] - outs[1][4]).max() <= 2e-3   # (y in fp16: one ulp where the statistics differ in their last bit)
    # configurations that cannot carry the statistics say so instead of silently skipping them
    d2 = _lib.ConvDesc.from_buffer_copy(d); d2.cfg = 2
    assert lib.y5_conv2d_fwd_stats(C.byref(d2), ptr(xa), ptr(wa), ptr(ba), ptr(z1), ptr(part), part.nbytes, C.byref(rows), None) == _lib.Y5_ERR_UNSUPPORTED
