"""CPU: the fused multi-tensor optimizer kernels (csrc/optim.hip, host-compiled on the HIP emulator) against the reference's
own step sequence (train.py:413-421): GradScaler.unscale_ -> clip_grad_norm_(10.0) -> torch.optim.SGD(momentum, nesterov)
with the three smart_optimizer groups -> ModelEMA.update (utils/torch_utils.py:257-290, 343-369)."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

from tests.hipemu.emu import emu
from yolov5_amd.torch_utils import HipSGD, ModelEMA, smart_optimizer


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 24, 3, bias=False)
        self.b1 = nn.BatchNorm2d(24)
        self.c2 = nn.Conv2d(24, 40, 5)            # 24000 weights: two 16384-element chunks
        self.b2 = nn.BatchNorm2d(40)
        self.head = nn.Conv2d(40, 7, 1)


def _grads(model, seed, scale=1.0, big=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for p in model.parameters():
        t = torch.randn(p.shape, generator=g) * (30.0 if big else 0.02) * scale
        out.append(t)
    return out


def _ref_optimizer(model):
    g = [], [], []
    for v in model.modules():
        for n, p in v.named_parameters(recurse=0):
            (g[2] if n == "bias" else g[1] if isinstance(v, nn.BatchNorm2d) else g[0]).append(p)
    opt = torch.optim.SGD(g[2], lr=0.01, momentum=0.937, nesterov=True)
    opt.add_param_group({"params": g[0], "weight_decay": 5e-4})
    opt.add_param_group({"params": g[1], "weight_decay": 0.0})
    return opt


def _set_lrs(opt, step):
    for i, g in enumerate(opt.param_groups):  # warm-up style: every group on its own schedule
        g["lr"] = 0.01 * (1.0 + 0.3 * i) / (1 + step)


@pytest.mark.parametrize("mode", ["plain", "scaled_clipped", "scaled_not_clipped"])
def test_fused_sgd_matches_torch_sequence(mode):
    torch.manual_seed(0)
    m_ref = Net()
    m_hip = copy.deepcopy(m_ref)
    opt_ref = _ref_optimizer(m_ref)
    opt_hip = smart_optimizer(m_hip, "SGD", lr=0.01, momentum=0.937, decay=5e-4)
    assert isinstance(opt_hip, HipSGD) and [len(g["params"]) for g in opt_hip.param_groups] == [len(g["params"]) for g in opt_ref.param_groups]
    opt_hip._lib = emu()
    S = 1.0 if mode == "plain" else 1024.0
    big = mode == "scaled_clipped"
    for step in range(4):
        _set_lrs(opt_ref, step)
        _set_lrs(opt_hip, step)
        gs = _grads(m_ref, 10 + step, S, big)
        for p, q, g in zip(m_ref.parameters(), m_hip.parameters(), gs):
            p.grad = g.clone()
            q.grad = g.clone()
        # reference sequence
        if mode != "plain":
            for p in m_ref.parameters():
                p.grad.mul_(1.0 / S)
            norm_ref = torch.nn.utils.clip_grad_norm_(m_ref.parameters(), max_norm=10.0)
        opt_ref.step()
        stats = opt_hip.step_fused(inv_scale=1.0 / S, max_norm=10.0) if mode != "plain" else opt_hip.step()
        if mode != "plain":
            np.testing.assert_allclose(float(stats[0]), float(norm_ref), rtol=2e-6)
            assert float(stats[2]) == 0.0
            assert (float(stats[1]) < 1.0) == big
        for (n, p), q in zip(m_ref.named_parameters(), m_hip.parameters()):
            torch.testing.assert_close(q, p, rtol=2e-6, atol=1e-7, msg=lambda s: f"{mode} step {step} {n}: {s}")
            torch.testing.assert_close(opt_hip.state[q]["momentum_buffer"], opt_ref.state[p]["momentum_buffer"], rtol=2e-6, atol=1e-7)
    # state_dict layout is torch.optim.SGD's
    sd = opt_hip.state_dict()
    assert set(sd["state"][0]) == {"momentum_buffer"} and sd["param_groups"][1]["weight_decay"] == 5e-4


def test_non_finite_gradient_skips_the_update_but_not_the_ema():
    torch.manual_seed(1)
    m = Net()
    opt = smart_optimizer(m, "SGD", lr=0.01, momentum=0.937, decay=5e-4)
    opt._lib = emu()
    ema = ModelEMA(m, tau=5)
    for p, g in zip(m.parameters(), _grads(m, 3, 256.0)):
        p.grad = g
    opt.step_fused(inv_scale=1 / 256.0, max_norm=10.0, ema=ema, model=m)
    before = [p.detach().clone() for p in m.parameters()]
    mom = [opt.state[p]["momentum_buffer"].clone() for p in m.parameters()]
    ema_before = [p.detach().clone() for p in ema.ema.parameters()]
    for p, g in zip(m.parameters(), _grads(m, 4, 256.0)):
        p.grad = g
    m.c2.weight.grad[1, 2, 0, 0] = float("inf")
    stats = opt.step_fused(inv_scale=1 / 256.0, max_norm=10.0, ema=ema, model=m)
    assert float(stats[2]) == 1.0
    for p, b, mb in zip(m.parameters(), before, mom):
        assert torch.equal(p, b) and torch.equal(opt.state[p]["momentum_buffer"], mb)
    d = ema.decay(2)
    for e, eb, p in zip(ema.ema.parameters(), ema_before, m.parameters()):  # train.py:417-421: ema.update runs even on a skipped step
        torch.testing.assert_close(e, eb * d + (1 - d) * p.detach(), rtol=1e-6, atol=1e-7)
    m.c1.weight.grad[0, 0, 0, 0] = float("nan")
    assert float(opt.step_fused(inv_scale=1 / 256.0, max_norm=10.0)[2]) == 1.0


def test_fused_ema_matches_model_ema_update():
    torch.manual_seed(2)
    m_ref, x = Net(), torch.randn(4, 3, 12, 12)
    m_hip = copy.deepcopy(m_ref)
    ema_ref, ema_hip = ModelEMA(m_ref, tau=3), ModelEMA(m_hip, tau=3)
    opt_ref = _ref_optimizer(m_ref)
    opt_hip = smart_optimizer(m_hip, "SGD", lr=0.01, momentum=0.937, decay=5e-4)
    opt_hip._lib = emu()
    for step in range(3):
        for mm in (m_ref, m_hip):   # move the BatchNorm running statistics (float buffers are part of the EMA)
            mm.train()
            mm.b1(mm.c1(x * (step + 1)))
        gs = _grads(m_ref, 20 + step)
        for p, q, g in zip(m_ref.parameters(), m_hip.parameters(), gs):
            p.grad, q.grad = g.clone(), g.clone()
        opt_ref.step()
        ema_ref.update(m_ref)
        opt_hip.step_fused(ema=ema_hip, model=m_hip)
        assert ema_hip.updates == ema_ref.updates
        for (k, a), b in zip(ema_ref.ema.state_dict().items(), ema_hip.ema.state_dict().values()):
            if a.dtype.is_floating_point:
                torch.testing.assert_close(b, a, rtol=2e-6, atol=1e-7, msg=lambda s: f"step {step} {k}: {s}")
            else:
                assert torch.equal(a, b)


def test_fused_step_and_ema_vs_reference_golden():
    """tests/golden/optim.npz was produced by the REFERENCE's smart_optimizer + ModelEMA (oracle/make_golden.py: gen_optim) on
    yolov5n: three steps of unscale -> clip_grad_norm_(10.0) -> SGD step -> ema.update.  Same weights, same synthetic gradients
    here through HipSGD.step_fused (kernel sources on the emulator)."""
    import os

    from oracle import yolo_oracle as yo
    from oracle.make_golden import optim_grad
    from yolov5_amd.yolo import DetectionModel

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim.npz"))
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False))
    m.train()
    opt = smart_optimizer(m, "SGD", 0.01, 0.937, 5e-4)
    opt._lib = emu()
    assert [len(x["params"]) for x in opt.param_groups] == g["group_sizes"].tolist()
    np.testing.assert_allclose([x["weight_decay"] for x in opt.param_groups], g["group_decay"])
    ema = ModelEMA(m, tau=4)
    S = 512.0
    for step in range(3):
        for i, x in enumerate(opt.param_groups):
            x["lr"] = 0.01 * (1.0 + 0.5 * i) / (1 + step)
        for k, p in m.named_parameters():
            p.grad = torch.from_numpy(optim_grad(k, p.shape, step)) * S
        stats = opt.step_fused(inv_scale=1.0 / S, max_norm=10.0, ema=ema, model=m)
        np.testing.assert_allclose(float(stats[0]), g["norms"][step], rtol=1e-5)
        assert float(stats[1]) < 1.0 and float(stats[2]) == 0.0
    esd = ema.ema.state_dict()
    n = 0
    for k, p in m.state_dict().items():
        if not p.dtype.is_floating_point:
            continue
        a, e = p.detach().numpy().ravel(), esd[k].detach().float().numpy().ravel()
        for got, want, tag in ((a, g["p:" + k], "param"), (e, g["e:" + k], "ema")):
            np.testing.assert_allclose(got[::97], want[:-1], rtol=2e-5, atol=2e-7, err_msg=f"{tag} {k}")
            np.testing.assert_allclose(got.astype(np.float64).sum(), want[-1], rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(got).sum())), err_msg=f"{tag} sum {k}")
        n += 1
    assert n > 200
