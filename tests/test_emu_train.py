"""CPU: the whole training plan (train-mode forward with batch-statistics BN, full backward to every parameter) run on
the HIP emulator for yolov5n @128x128, against torch autograd over the CPU oracle (reference semantics of model.train(),
models/common.py:82-88 + models/yolo.py:96-98)."""
import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo
from tests.hipemu.backend import EmuBackend
from yolov5_amd.train_engine import TrainEngine
from yolov5_amd.yolo import DetectionModel


def test_yolov5n_train_forward_backward_vs_oracle_autograd():
    cfg = yo.model_cfg("yolov5n")
    sd = yo.det_state_dict(cfg, 0, fused=False)
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(sd)
    m.train()
    B = 2
    x = torch.from_numpy(detgen.uniform((B, 3, 128, 128), 0.0, 1.0, name="img", seed=0))
    eng = TrainEngine(m, (B, 3, 128, 128), "cpu", backend=EmuBackend())
    p = [eng.be.to_torch(o).float() for o in eng.forward(x.half())]

    # oracle: same weights as leaf tensors, train-mode BN
    sdo = {k: v.clone() for k, v in sd.items()}
    leaves = {}
    for k, v in sdo.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var", "anchors")):
            sdo[k] = v.clone().requires_grad_(True)
            leaves[k] = sdo[k]
    ref = yo.model_forward(cfg, sdo, x, training=True, bn_batch_stats=True)
    for a, b in zip(p, ref):
        assert a.shape == b.shape
        scale = float(b.detach().abs().max())
        # fp16 activations vs the fp32 oracle through 60 batch-normalised layers (the deepest have 32 samples per channel)
        assert float((a - b.detach()).abs().max()) < 6e-2 * max(scale, 1.0)
        assert float((a - b.detach()).abs().mean()) < 1e-2 * max(scale, 1.0)
    # running statistics were updated like torch's (momentum 0.03)
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            np.testing.assert_allclose(mod.running_mean.numpy(), sdo[name + ".running_mean"].numpy(), rtol=2e-2, atol=2e-3)
            np.testing.assert_allclose(mod.running_var.numpy(), sdo[name + ".running_var"].numpy(), rtol=2e-2, atol=2e-3)
            assert int(mod.num_batches_tracked) == 1
    # backward with a fixed upstream gradient
    rs = [torch.from_numpy(detgen.uniform(tuple(b.shape), -1, 1, name=f"up{i}", seed=3)) for i, b in enumerate(ref)]
    sum((b * r).sum() for b, r in zip(ref, rs)).backward()
    grads = eng.backward([r.half() for r in rs])
    names = [n for n, _ in m.named_parameters()]
    # fp16 activations + batch statistics over as few as 32 samples make individual elements noisy against the fp32
    # oracle; direction and norm of every parameter gradient must agree (the kernels are unit-tested to 1e-3 each)
    worst_cos, worst_rel = 1.0, 0.0
    for n, g in zip(names, grads):
        assert g is not None, n
        rg = leaves[n].grad
        assert rg is not None and g.shape == rg.shape, n
        a, b = g.float().flatten().double(), rg.flatten().double()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        worst_cos, worst_rel = min(worst_cos, cos), max(worst_rel, rel)
        assert cos > 0.97 and rel < 0.25, (n, cos, rel)
    print(f"\n[train-emu] {len(names)} parameter gradients: worst cosine {worst_cos:.4f}, worst relative L2 error {worst_rel:.4f}")
    assert len(names) == len(list(m.parameters()))


def test_filter_jobs_match_single_filter_entry_points():
    """y5_filter_jobs (one launch over many filters) against y5_pack_conv_weight / y5_pack_dgrad_weight / y5_unpack_conv_wgrad."""
    import ctypes as C

    import numpy as np

    from tests.hipemu.emu import aligned, emu
    from yolov5_amd import _lib
    from yolov5_amd.packing import round_up

    lib = emu()
    rng = np.random.default_rng(0)
    shapes = [(40, 24, 3, 3, 24), (255, 64, 1, 1, 64), (16, 3, 6, 6, 4), (72, 48, 3, 3, 48)]
    jobs, checks, keep = [], [], []
    for c2, c1, kh, kw, c1v in shapes:
        w = aligned((c2, c1, kh, kw), np.float32)
        w[...] = rng.standard_normal(w.shape).astype(np.float32)
        Kpad, Npad = round_up(kh * kw * c1v, 64), round_up(c2, 32)
        # kind 0
        ref = aligned((Npad, Kpad), np.float16, 7)
        assert lib.y5_pack_conv_weight(C.c_void_p(w.ctypes.data), c2, c1, kh, kw, c1v, C.c_void_p(ref.ctypes.data), Kpad, Npad, None) == 0
        out = aligned((Npad, Kpad), np.float16, 9)
        jobs.append((w, out, Npad * Kpad, 0, c2, c1, kh, kw, c1v, 0, Kpad, Npad, 0, 0, (), ()))
        checks.append((out, ref))
        # kind 2 (unpack of a packed fp32 gradient)
        dw = aligned((Npad, Kpad), np.float32)
        dw[...] = rng.standard_normal(dw.shape).astype(np.float32)
        gref = aligned((c2, c1, kh, kw), np.float32, 1)
        assert lib.y5_unpack_conv_wgrad(C.c_void_p(dw.ctypes.data), Kpad, C.c_void_p(gref.ctypes.data), c2, c1, kh, kw, c1v, None) == 0
        gout = aligned((c2, c1, kh, kw), np.float32, 2)
        jobs.append((dw, gout, c2 * c1 * kh * kw, 2, c2, c1, kh, kw, c1v, 0, Kpad, Npad, 0, 0, (), ()))
        checks.append((gout, gref))
        # kind 1 (data-gradient sub-filter: taps (0, 2) x (1,) where they exist)
        th, tw = ((0, 2), (1,)) if kh >= 3 else ((0,), (0,))
        c2v = round_up(c2, 8)
        Kp2, Np2 = round_up(len(th) * len(tw) * c2v, 64), round_up(c1, 32)
        dref = aligned((Np2, Kp2), np.float16, 3)
        assert lib.y5_pack_dgrad_weight(C.c_void_p(w.ctypes.data), c2, c1, kh, kw, (C.c_int * len(th))(*th), len(th), (C.c_int * len(tw))(*tw), len(tw),
                                        c2v, C.c_void_p(dref.ctypes.data), Kp2, Np2, None) == 0
        dout = aligned((Np2, Kp2), np.float16, 4)
        jobs.append((w, dout, Np2 * Kp2, 1, c2, c1, kh, kw, 0, c2v, Kp2, Np2, len(th), len(tw), th, tw))
        checks.append((dout, dref))
        keep += [w, dw]
    arr = (_lib.FilterJob * len(jobs))()
    for j, r in zip(arr, jobs):
        j.src, j.dst = r[0].ctypes.data, r[1].ctypes.data
        (j.total, j.kind, j.C2, j.C1, j.KH, j.KW, j.C1_view, j.C2_view, j.Kpad, j.Npad, j.nth, j.ntw) = r[2:14]
        for q, v in enumerate(r[14]):
            j.th[q] = v
        for q, v in enumerate(r[15]):
            j.tw[q] = v
    assert lib.y5_filter_jobs(C.byref(arr), len(jobs), max(r[2] for r in jobs), None) == 0, lib.y5_last_error()
    for got, want in checks:
        assert np.array_equal(got, want)


def test_fp32_training_plan_exact_gradients_vs_oracle_autograd():
    """TrainEngine(dtype=float32): the whole step in fp32 (exact-fp32 MFMA forward / data gradient, plain fp32 weight gradient, fp32
    BatchNorm / SiLU / pooling / upsample backward) against torch autograd over the CPU oracle -- no fp16 envelope: every parameter
    gradient to 1e-3 relative L2 (the tiny model of oracle/make_golden.py:TINY_CFG, 2 x 3 x 64 x 64)."""
    import copy

    from oracle.make_golden import TINY_CFG

    cfg = copy.deepcopy(TINY_CFG)
    sd = yo.det_state_dict(cfg, 4, fused=False)
    m = DetectionModel(copy.deepcopy(TINY_CFG))
    m.load_state_dict(sd)
    m.train()
    B = 2
    x = torch.from_numpy(detgen.uniform((B, 3, 64, 64), 0.0, 1.0, name="img", seed=4))
    eng = TrainEngine(m, (B, 3, 64, 64), "cpu", backend=EmuBackend(), dtype=torch.float32)
    p = [eng.be.to_torch(o) for o in eng.forward(x)]
    assert all(t.dtype == torch.float32 for t in p)
    sdo, leaves = {}, {}
    for k, v in sd.items():
        sdo[k] = v.clone()
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var", "anchors")):
            sdo[k] = v.clone().requires_grad_(True)
            leaves[k] = sdo[k]
    ref = yo.model_forward(cfg, sdo, x, training=True, bn_batch_stats=True)
    for a, b in zip(p, ref):
        np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=2e-4, atol=2e-4)
    rs = [torch.from_numpy(detgen.uniform(tuple(b.shape), -1, 1, name=f"up{i}", seed=5)) for i, b in enumerate(ref)]
    sum((b * r).sum() for b, r in zip(ref, rs)).backward()
    grads = eng.backward(rs)
    worst = 0.0
    for (n, _), g in zip(m.named_parameters(), grads):
        rg = leaves[n].grad
        rel = float((g.double() - rg.double()).norm() / (rg.double().norm() + 1e-30))
        worst = max(worst, rel)
        assert rel < 1e-3, (n, rel)
    print(f"\\n[train-emu fp32] worst relative L2 error over {len(grads)} parameter gradients: {worst:.2e}")
