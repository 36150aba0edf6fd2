"""GPU: fused optimizer step (csrc/optim.hip) against torch's own sequence on the same device -- train.py:413-421."""
import copy

import numpy as np
import pytest
import torch

from tests.test_emu_optim import Net, _grads, _ref_optimizer, _set_lrs
from yolov5_amd.torch_utils import HipSGD, ModelEMA, smart_optimizer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("mode", ["plain", "scaled_clipped"])
def test_fused_step_matches_torch_on_device(mode, dev):
    torch.manual_seed(0)
    m_ref = Net().to(dev)
    m_hip = copy.deepcopy(m_ref)
    ema_ref, ema_hip = ModelEMA(m_ref, tau=3), ModelEMA(m_hip, tau=3)
    opt_ref = _ref_optimizer(m_ref)
    opt_hip = smart_optimizer(m_hip, "SGD", lr=0.01, momentum=0.937, decay=5e-4)
    assert isinstance(opt_hip, HipSGD)
    S, big = (1.0, False) if mode == "plain" else (4096.0, True)
    for step in range(5):
        _set_lrs(opt_ref, step)
        _set_lrs(opt_hip, step)
        gs = _grads(m_ref, 10 + step, S, big)
        for p, q, g in zip(m_ref.parameters(), m_hip.parameters(), gs):
            p.grad, q.grad = g.to(dev), g.to(dev)
        if mode != "plain":
            for p in m_ref.parameters():
                p.grad.mul_(1.0 / S)
            norm_ref = torch.nn.utils.clip_grad_norm_(m_ref.parameters(), max_norm=10.0)
        opt_ref.step()
        ema_ref.update(m_ref)
        stats = opt_hip.step_fused(inv_scale=1.0 / S, max_norm=10.0 if mode != "plain" else 0.0, ema=ema_hip, model=m_hip)
        if mode != "plain":
            np.testing.assert_allclose(float(stats[0]), float(norm_ref), rtol=5e-6)
            assert float(stats[1]) < 1.0 and float(stats[2]) == 0.0
        for (n, p), q in zip(m_ref.named_parameters(), m_hip.parameters()):
            torch.testing.assert_close(q, p, rtol=3e-6, atol=2e-7, msg=lambda s: f"{mode} step {step} {n}: {s}")
        for (k, a), b in zip(ema_ref.ema.state_dict().items(), ema_hip.ema.state_dict().values()):
            if a.dtype.is_floating_point:
                torch.testing.assert_close(b, a, rtol=3e-6, atol=2e-7, msg=lambda s: f"ema {step} {k}: {s}")


def test_non_finite_skip_on_device(dev):
    torch.manual_seed(1)
    m = Net().to(dev)
    opt = smart_optimizer(m, "SGD", lr=0.01, momentum=0.937, decay=5e-4)
    for p, g in zip(m.parameters(), _grads(m, 3, 256.0)):
        p.grad = g.to(dev)
    opt.step_fused(inv_scale=1 / 256.0, max_norm=10.0)
    before = [p.detach().clone() for p in m.parameters()]
    m.c2.weight.grad[1, 2, 0, 0] = float("inf")
    stats = opt.step_fused(inv_scale=1 / 256.0, max_norm=10.0)
    assert float(stats[2]) == 1.0
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b)


def test_fused_step_and_ema_vs_reference_golden_on_gpu(dev):
    """tests/golden/optim.npz -- parameters and EMA tensors after three steps of the REFERENCE's own smart_optimizer
    (torch SGD-Nesterov, three groups, train.py:413-421) + clip_grad_norm_(10.0) + ModelEMA.update (utils/torch_utils.py:343-369) on yolov5n,
    made by oracle/make_golden.py:gen_optim from the unmodified reference -- replayed through the fused kernels ON THE GPU (round 2 replayed
    it on the emulator only; the GPU test above compares with torch SGD and the product's own EMA host path)."""
    import os

    from oracle import yolo_oracle as yo
    from oracle.make_golden import optim_grad
    from yolov5_amd.yolo import DetectionModel

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim.npz"))
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False))
    m = m.to(dev).train()
    opt = smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4)
    assert isinstance(opt, HipSGD)
    assert [len(x["params"]) for x in opt.param_groups] == g["group_sizes"].tolist()
    np.testing.assert_allclose([x["weight_decay"] for x in opt.param_groups], g["group_decay"])
    ema = ModelEMA(m, tau=4)
    S = 512.0
    for step in range(3):
        for i, x in enumerate(opt.param_groups):
            x["lr"] = 0.01 * (1.0 + 0.5 * i) / (1 + step)
        for k, p in m.named_parameters():
            p.grad = (torch.from_numpy(optim_grad(k, p.shape, step)) * S).to(dev)
        stats = opt.step_fused(inv_scale=1.0 / S, max_norm=10.0, ema=ema, model=m)
        np.testing.assert_allclose(float(stats[0]), g["norms"][step], rtol=1e-5)
        assert float(stats[1]) < 1.0 and float(stats[2]) == 0.0
    esd = ema.ema.state_dict()
    n = 0
    for k, p in m.state_dict().items():
        if not p.dtype.is_floating_point:
            continue
        a, e = p.detach().float().cpu().numpy().ravel(), esd[k].detach().float().cpu().numpy().ravel()
        for got, want, tag in ((a, g["p:" + k], "param"), (e, g["e:" + k], "ema")):
            np.testing.assert_allclose(got[::97], want[:-1], rtol=2e-5, atol=2e-7, err_msg=f"{tag} {k}")
            np.testing.assert_allclose(got.astype(np.float64).sum(), want[-1], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(got).sum())), err_msg=f"{tag} sum {k}")
        n += 1
    assert n > 200
