"""GPU parity of yolov5_amd.loss.ComputeLoss (HIP: y5_loss_forward / y5_loss_backward through the C-ABI) against the
reference-generated golden fixtures and, at BASELINE size (64 x 25200 cells, 512 targets), against the CPU oracle."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo
from oracle.make_golden import loss_case

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss.npz"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def compute_loss(dev):
    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.yolo import DetectionModel

    m = DetectionModel("yolov5s.yaml").to(dev)
    m.hyp = dict(yo.HYP_SCRATCH_LOW)
    assert np.array_equal(m.model[-1].anchors.cpu().numpy(), G["anchors"])
    return ComputeLoss(m)


@pytest.mark.parametrize("name", ["appendix_a", "synthetic", "no_targets"])
def test_loss_vs_reference_golden(name, compute_loss, dev):
    pn, tn = loss_case(name)
    p = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in pn]
    t = torch.from_numpy(tn).to(dev)
    tcls, tbox, indices, anch = compute_loss.build_targets(p, t)
    for i in range(3):
        idx = torch.stack(indices[i]).cpu().numpy()
        assert idx.dtype == np.int64 and np.array_equal(idx, G[f"{name}_idx{i}"])  # bit-exact, reference row order
        assert np.array_equal(tcls[i].cpu().numpy(), G[f"{name}_tcls{i}"])
        assert np.array_equal(tbox[i].cpu().numpy(), G[f"{name}_tbox{i}"])
        assert np.array_equal(anch[i].cpu().numpy(), G[f"{name}_anch{i}"])
    loss, items = compute_loss(p, t)
    assert loss.shape == (1,) and items.shape == (3,) and not items.requires_grad
    loss.backward()
    np.testing.assert_allclose(loss.item(), G[f"{name}_loss"][0], rtol=1e-4)       # north_star: fp32 loss within 1e-4
    np.testing.assert_allclose(items.cpu().numpy(), G[f"{name}_items"], rtol=1e-4, atol=1e-6)
    for i in range(3):
        g = p[i].grad.cpu().numpy()
        if f"{name}_grad{i}" in G:
            np.testing.assert_allclose(g, G[f"{name}_grad{i}"], rtol=1e-3, atol=1e-8)
        else:
            s = g.astype(np.float64)
            np.testing.assert_allclose([s.sum(), np.abs(s).sum()], G[f"{name}_grad{i}_sum"], rtol=1e-4)
            nz = G[f"{name}_grad{i}_nzidx"]
            if len(nz):
                np.testing.assert_allclose(g[tuple(nz.T)], G[f"{name}_grad{i}_nzrows"], rtol=1e-3, atol=1e-8)


def test_loss_full_size_vs_oracle_and_determinism(compute_loss, dev):
    """bs=64, 640^2 (3 x 64 x 25200/3 cells, 85 outputs), 512 targets: loss/items vs the CPU oracle within 1e-4,
    gradient checksums, run-to-run bit-identical results (deterministic reductions), scaled fp16 backward."""
    bs = 64
    pn = [detgen.uniform((bs, 3, s, s, 85), -4.0, 2.0, name=f"L{s}", seed=21) for s in (80, 40, 20)]
    tn = detgen.synth_targets(bs, 8, seed=21)
    anchors = torch.from_numpy(G["anchors"])
    pc = [torch.from_numpy(a).requires_grad_(True) for a in pn]
    t0 = time.time()
    ref_loss, ref_items = yo.compute_loss(pc, torch.from_numpy(tn), anchors)
    ref_loss.backward()
    cpu_s = time.time() - t0
    p = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in pn]
    t = torch.from_numpy(tn).to(dev)
    loss, items = compute_loss(p, t)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=1e-4)
    np.testing.assert_allclose(items.cpu().numpy(), ref_items.numpy(), rtol=1e-4)
    for i in range(3):
        g, r = p[i].grad.cpu().numpy().astype(np.float64), pc[i].grad.numpy().astype(np.float64)
        np.testing.assert_allclose([g.sum(), np.abs(g).sum()], [r.sum(), np.abs(r).sum()], rtol=1e-4)
        np.testing.assert_allclose(g, r, rtol=2e-3, atol=1e-9)
    # determinism: a second evaluation gives bit-identical loss and gradients
    p2 = [q.detach().clone().requires_grad_(True) for q in p]
    loss2, _ = compute_loss(p2, t)
    loss2.backward()
    assert torch.equal(loss, loss2) and all(torch.equal(a.grad, b.grad) for a, b in zip(p, p2))
    # fp16 + GradScaler-style scaling
    ph = [q.detach().half().requires_grad_(True) for q in p]
    lh, _ = compute_loss(ph, t)
    (lh * 65536.0).backward()
    assert ph[0].grad.dtype == torch.float16
    np.testing.assert_allclose(lh.item(), ref_loss.item(), rtol=3e-3)
    gh = ph[0].grad.float().cpu().numpy() / 65536.0
    np.testing.assert_allclose(np.abs(gh).sum(), np.abs(pc[0].grad.numpy()).sum(), rtol=2e-2)
    # timing (informational): forward + backward on the GPU vs the CPU oracle
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        q = [x.detach().requires_grad_(True) for x in ph]
        l, _ = compute_loss(q, t)
        l.backward()
    torch.cuda.synchronize()
    print(f"\n[loss] bs=64 640^2 nt=512 fp16: {(time.time() - t0) / 10 * 1e3:.3f} ms fwd+bwd on MI355X; CPU oracle fp32 {cpu_s * 1e3:.0f} ms")


def test_loss_focal_vs_reference_golden(dev):
    """hyp fl_gamma = 1.5 + label smoothing 0.1 through yolov5_amd.loss.ComputeLoss on the GPU against the reference's FocalLoss-wrapped ComputeLoss."""
    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.yolo import DetectionModel

    F_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_focal.npz"))
    m = DetectionModel("yolov5s.yaml").to(dev)
    m.hyp = dict(yo.HYP_SCRATCH_LOW, fl_gamma=1.5, label_smoothing=0.1)
    cl = ComputeLoss(m)
    pn, tn = loss_case("synthetic")
    p = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in pn]
    loss, items = cl(p, torch.from_numpy(tn).to(dev))
    loss.backward()
    np.testing.assert_allclose(loss.item(), F_["loss"][0], rtol=1e-4)
    np.testing.assert_allclose(items.cpu().numpy(), F_["items"], rtol=1e-4)
    for i in range(3):
        g = p[i].grad.cpu().numpy()
        s = g.astype(np.float64)
        np.testing.assert_allclose([s.sum(), np.abs(s).sum()], F_[f"grad{i}_sum"], rtol=1e-4)
        nz = F_[f"grad{i}_nzidx"]
        np.testing.assert_allclose(g[tuple(nz.T)], F_[f"grad{i}_nzrows"], rtol=1e-3, atol=1e-8)
        np.testing.assert_allclose(g[0, 0, :4, :8, 4], F_[f"grad{i}_obj_head"], rtol=1e-3, atol=1e-9)


def test_loss_autobalance_vs_oracle(dev):
    """ComputeLoss(model, autobalance=True) on the GPU (utils/loss.py:127, :173-177): three calls, losses and the drifting balance list against the
    oracle (pinned to the live reference in tests/test_oracle_vs_reference.py::test_autobalance_matches_live_reference)."""
    from oracle import detgen
    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.yolo import DetectionModel

    m = DetectionModel("yolov5s.yaml").to(dev)
    m.hyp = dict(yo.HYP_SCRATCH_LOW)
    cl = ComputeLoss(m, autobalance=True)
    assert cl.ssi == 1
    anchors = yo.model_anchors(yo.model_cfg("yolov5s"))
    bal = [4.0, 1.0, 0.4]
    for step in range(3):
        pn = [detgen.uniform((2, 3, s, s, 85), -3.0, 3.0, name=f"ab{s}", seed=40 + step) for s in (16, 8, 4)]
        tn = detgen.synth_targets(2, 6, seed=40 + step)
        p = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in pn]
        q = [torch.from_numpy(a).clone().requires_grad_(True) for a in pn]
        loss, items = cl(p, torch.from_numpy(tn).to(dev))
        loss.backward()
        lo, io = yo.compute_loss(q, torch.from_numpy(tn), anchors, nc=80, balance=bal, autobalance_ssi=1)
        lo.backward()
        np.testing.assert_allclose(loss.item(), lo.item(), rtol=1e-5)
        np.testing.assert_allclose(items.cpu().numpy(), io.numpy(), rtol=1e-5)
        np.testing.assert_allclose(cl.balance, bal, rtol=1e-8)
        for a, b in zip(p, q):
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-3, atol=1e-8)
