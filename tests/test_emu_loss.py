"""CPU: the product's ComputeLoss kernels (yolov5_amd/csrc/loss_kernels.h) compiled for the host on the HIP emulator,
checked against the reference-generated golden fixtures (tests/golden/loss.npz) and the oracle's autograd."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import yolo_oracle as yo
from oracle.make_golden import loss_case
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib

import os

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss.npz"))
HYP = yo.HYP_SCRATCH_LOW


def make_desc(shapes, anchors, dtype, fl_gamma=0.0, smoothing=0.0):
    d = _lib.LossDesc()
    d.dtype = _lib.Y5_F16 if dtype == np.float16 else _lib.Y5_F32
    d.nl, d.na, d.nc, d.bs = len(shapes), shapes[0][1], shapes[0][4] - 5, shapes[0][0]
    for i, s in enumerate(shapes):
        d.ny[i], d.nx[i] = s[2], s[3]
        d.balance[i] = [4.0, 1.0, 0.4][i]
        for a in range(d.na):
            d.anchors[i * 16 + a * 2], d.anchors[i * 16 + a * 2 + 1] = float(anchors[i, a, 0]), float(anchors[i, a, 1])
    d.hyp_box, d.hyp_obj, d.hyp_cls = HYP["box"], HYP["obj"], HYP["cls"]
    d.cls_pw, d.obj_pw, d.anchor_t, d.cp, d.cn = HYP["cls_pw"], HYP["obj_pw"], HYP["anchor_t"], 1.0 - 0.5 * smoothing, 0.5 * smoothing
    d.fl_gamma = fl_gamma
    return d


def run_emu_loss(pn, tn, anchors, dtype=np.float32, scale=None, fl_gamma=0.0, smoothing=0.0):
    lib = emu()
    d = make_desc([p.shape for p in pn], anchors, dtype, fl_gamma, smoothing)
    nt = len(tn)
    nbytes = lib.y5_loss_workspace_bytes(C.byref(d), nt)
    assert nbytes > 0
    ws = aligned((nbytes,), np.uint8)
    P = []
    for p in pn:
        a = aligned(p.shape, dtype)
        a[...] = p.astype(dtype)
        P.append(a)
    t = aligned((max(nt, 1), 6), np.float32)
    t[:nt] = tn
    out = aligned((4,), np.float32)
    pp = (C.c_void_p * len(P))(*[a.ctypes.data for a in P])
    rc = lib.y5_loss_forward(C.byref(d), pp, ptr(t), nt, ptr(out), ptr(ws), nbytes, None)
    assert rc == 0, lib.y5_last_error()
    D = [aligned(p.shape, dtype, 7) for p in pn]
    dd = (C.c_void_p * len(D))(*[a.ctypes.data for a in D])
    gs = None
    if scale is not None:
        gs = aligned((1,), np.float32, scale)
    rc = lib.y5_loss_backward(C.byref(d), pp, nt, ptr(gs), dd, ptr(ws), nbytes, None)
    assert rc == 0, lib.y5_last_error()
    rows = []
    offs = (C.c_size_t * 10)()
    cap = C.c_longlong(0)
    for i in range(d.nl):
        assert lib.y5_loss_targets_layout(C.byref(d), nt, i, offs, C.byref(cap)) == 0
        n = int(ws[offs[0]:offs[0] + 4].view(np.int32)[0])
        ints = [ws[offs[k]:offs[k] + 4 * n].view(np.int32).copy() for k in (1, 2, 3, 4, 5)]
        tb = ws[offs[6]:offs[6] + 16 * n].view(np.float32).reshape(n, 4).copy()
        an = ws[offs[7]:offs[7] + 8 * n].view(np.float32).reshape(n, 2).copy()
        rows.append((n, ints, tb, an))
    return out.copy(), D, rows


@pytest.mark.parametrize("name", ["appendix_a", "synthetic", "no_targets"])
def test_emu_loss_vs_reference_golden(name):
    pn, tn = loss_case(name)
    out, D, rows = run_emu_loss(pn, tn, G["anchors"])
    for i, (n, (b, a, gj, gi, c), tb, an) in enumerate(rows):
        ref = G[f"{name}_idx{i}"]
        assert n == ref.shape[1]
        assert np.array_equal(np.stack([b, a, gj, gi]).astype(np.int64), ref)  # bit-exact indices, reference row order
        assert np.array_equal(c.astype(np.int64), G[f"{name}_tcls{i}"])
        assert np.array_equal(tb, G[f"{name}_tbox{i}"])
        assert np.array_equal(an, G[f"{name}_anch{i}"])
    np.testing.assert_allclose(out[0], G[f"{name}_loss"][0], rtol=1e-5)
    np.testing.assert_allclose(out[1:], G[f"{name}_items"], rtol=1e-5, atol=1e-7)
    for i in range(3):
        if f"{name}_grad{i}" in G:
            np.testing.assert_allclose(D[i], G[f"{name}_grad{i}"], rtol=2e-4, atol=2e-9)
        else:
            s = D[i].astype(np.float64)
            np.testing.assert_allclose([s.sum(), np.abs(s).sum()], G[f"{name}_grad{i}_sum"], rtol=1e-5)
            nz = G[f"{name}_grad{i}_nzidx"]
            if len(nz):
                np.testing.assert_allclose(D[i][tuple(nz.T)], G[f"{name}_grad{i}_nzrows"], rtol=2e-4, atol=2e-9)


def test_emu_loss_grad_scale_and_fp16():
    """grad_scale multiplies the gradient before the cast to p's dtype (GradScaler semantics); fp16 logits are
    widened to fp32 and tobj is rounded to fp16 like the reference's `iou.type(tobj.dtype)` (loss.py:157)."""
    pn, tn = loss_case("synthetic")
    ph = [p.astype(np.float16) for p in pn]
    out, D, _ = run_emu_loss(ph, tn, G["anchors"], dtype=np.float16, scale=1024.0)
    anchors = torch.from_numpy(G["anchors"])
    p = [torch.from_numpy(a.astype(np.float32)).requires_grad_(True) for a in ph]
    loss, items = yo.compute_loss(p, torch.from_numpy(tn), anchors)
    (loss * 1024.0).backward()
    np.testing.assert_allclose(out[0], loss.item(), rtol=2e-4)
    for i in range(3):
        ref = p[i].grad.numpy()
        got = D[i].astype(np.float32)
        assert got.dtype == np.float32 and D[i].dtype == np.float16
        # matched cells: tobj is rounded to fp16 (|dt| <= 2.5e-4) times gobj*scale = 1.33 -> atol 4e-4; everything
        # else is one fp16 rounding of the scaled gradient
        np.testing.assert_allclose(got, ref, rtol=4e-3, atol=4e-4)
        unmatched = np.abs(ref[..., :4]).sum(-1) == 0
        np.testing.assert_allclose(got[unmatched], ref[unmatched], rtol=4e-3, atol=6e-6)


def test_emu_loss_duplicate_cells_last_write_wins():
    """Two targets in the same cell with the same anchor set: tobj takes the later row's iou and both rows' gradients
    are summed (SURVEY 8c hazard 3)."""
    rng = np.random.default_rng(3)
    pn = [rng.uniform(-2, 2, (1, 3, s, s, 9)).astype(np.float32) for s in (8, 4, 2)]
    tn = np.array([[0, 1, 0.52, 0.52, 0.2, 0.25], [0, 2, 0.53, 0.53, 0.22, 0.2], [0, 3, 0.52, 0.52, 0.3, 0.3]], np.float32)
    out, D, rows = run_emu_loss(pn, tn, G["anchors"])
    anchors = torch.from_numpy(G["anchors"])
    p = [torch.from_numpy(a).requires_grad_(True) for a in pn]
    loss, items = yo.compute_loss(p, torch.from_numpy(tn), anchors, nc=4)
    loss.backward()
    np.testing.assert_allclose(out[0], loss.item(), rtol=1e-5)
    np.testing.assert_allclose(out[1:], items.numpy(), rtol=1e-5, atol=1e-7)
    for i in range(3):
        np.testing.assert_allclose(D[i], p[i].grad.numpy(), rtol=2e-4, atol=2e-9)


def test_emu_loss_focal_vs_reference_golden():
    """fl_gamma = 1.5, label smoothing 0.1: the kernels' focal BCE (value and derivative) against the reference's FocalLoss-wrapped ComputeLoss."""
    F_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_focal.npz"))
    pn, tn = loss_case("synthetic")
    out, D, _rows = run_emu_loss(pn, tn, G["anchors"], fl_gamma=1.5, smoothing=0.1)
    np.testing.assert_allclose(out[0], F_["loss"][0], rtol=1e-5)
    np.testing.assert_allclose(out[1:], F_["items"], rtol=1e-5)
    for i in range(3):
        s = D[i].astype(np.float64)
        np.testing.assert_allclose([s.sum(), np.abs(s).sum()], F_[f"grad{i}_sum"], rtol=2e-5)
        nz = F_[f"grad{i}_nzidx"]
        np.testing.assert_allclose(D[i][tuple(nz.T)], F_[f"grad{i}_nzrows"], rtol=3e-4, atol=2e-9)
        np.testing.assert_allclose(D[i][0, 0, :4, :8, 4], F_[f"grad{i}_obj_head"], rtol=3e-4, atol=1e-10)
