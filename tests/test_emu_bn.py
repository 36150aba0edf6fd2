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
