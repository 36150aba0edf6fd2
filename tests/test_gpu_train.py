"""GPU: one training step through the reference-shaped API -- model.train(); pred = model(imgs);
loss, items = ComputeLoss(model)(pred, targets); (loss * scale).backward()  (train.py:401-410) -- with every kernel on the
MI355X, against torch autograd over the CPU oracle (fp32, train-mode BN)."""
import time

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(name, dev):
    from yolov5_amd.yolo import DetectionModel

    cfg = yo.model_cfg(name)
    sd = yo.det_state_dict(cfg, 0, fused=False)
    m = DetectionModel(name + ".yaml")
    m.load_state_dict(sd)
    m.hyp = dict(yo.HYP_SCRATCH_LOW)
    return m.to(dev).train(), cfg, sd


class _RoundSTE(torch.autograd.Function):
    """fp16 storage of a conv output with a straight-through gradient; mode 1 = round to nearest, mode > 1 = half-ulp dither."""

    @staticmethod
    def forward(ctx, y, mode):
        if mode > 1:
            g = torch.Generator().manual_seed(mode)
            y = y * (1 + (torch.rand(y.shape, generator=g) - 0.5) * 2.0 ** -10)
        return y.half().float()

    @staticmethod
    def backward(ctx, g):
        return g, None


def _oracle_grads(cfg, sd, x, t, mode):
    """torch autograd over the CPU oracle (fp32, train-mode BN); mode != 0 stores every conv output in fp16."""
    sdo, leaves = {}, {}
    for k, v in sd.items():
        sdo[k] = v.clone()
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var", "anchors")):
            sdo[k] = v.clone().requires_grad_(True)
            leaves[k] = sdo[k]
    orig = yo.F.conv2d
    if mode:
        yo.F.conv2d = lambda *a, **kw: _RoundSTE.apply(orig(*a, **kw), mode)
    try:
        ref = yo.model_forward(cfg, sdo, x, training=True, bn_batch_stats=True)
    finally:
        yo.F.conv2d = orig
    rloss, ritems = yo.compute_loss(ref, t, yo.model_anchors(cfg))
    rloss.backward()
    return sdo, leaves, ref, rloss.detach(), ritems.detach()


def test_train_step_matches_oracle_autograd(dev):
    from yolov5_amd.loss import ComputeLoss

    m, cfg, sd = _model("yolov5n", dev)
    B, S, SCALE = 8, 256, 4096.0
    x = torch.from_numpy(detgen.uniform((B, 3, S, S), 0.0, 1.0, name="timg", seed=7))
    t = torch.from_numpy(detgen.synth_targets(B, 6, seed=7))
    compute_loss = ComputeLoss(m)
    pred = m(x.half().to(dev))
    assert isinstance(pred, list) and len(pred) == 3 and pred[0].shape == (B, 3, S // 8, S // 8, 85) and pred[0].requires_grad
    loss, items = compute_loss(pred, t.to(dev))
    (loss * SCALE).backward()
    torch.cuda.synchronize()

    sdo, leaves, ref, rloss, ritems = _oracle_grads(cfg, sd, x, t, 0)
    for a, b in zip(pred, ref):
        d = (a.detach().float().cpu() - b.detach()).abs()
        assert float(d.max()) < 4e-2 * float(b.detach().abs().max()) and float(d.mean()) < 4e-3 * float(b.detach().abs().max())
    np.testing.assert_allclose(loss.item(), rloss.item(), rtol=2e-2)
    np.testing.assert_allclose(items.cpu().numpy(), ritems.numpy(), rtol=3e-2)
    # Noise floor of this comparison.  Gradients of a randomly initialised SiLU+BatchNorm stack are ill-conditioned at fp16
    # resolution: merely storing every conv output in fp16 (as the HIP path does) moves the ORACLE's own parameter gradients
    # by 5-20 % relative L2.  The envelope below is measured, per parameter, from three fp16-storage variants of the oracle
    # (round-to-nearest, and two half-ulp dithers); the HIP gradients must lie within 3x of it.  (The exact checks of the
    # backward kernels are the per-op tests of test_gpu_train_ops.py and the loss tests of test_gpu_loss.py.)
    env = {n: 0.0 for n in leaves}
    for mode in (1, 2, 3):
        lv = _oracle_grads(cfg, sd, x, t, mode)[1]
        for n in leaves:
            a, b = lv[n].grad.flatten().double(), leaves[n].grad.flatten().double()
            env[n] = max(env[n], float((a - b).norm() / (b.norm() + 1e-30)))
    rows = []
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        a, b = (p.grad.float().cpu().flatten() / SCALE).double(), leaves[n].grad.flatten().double()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        rows.append((cos, rel, env[n], n))
    worst_cos, worst_rel = min(r[0] for r in rows), max(r[1] for r in rows)
    med_rel, med_env = float(np.median([r[1] for r in rows])), float(np.median([r[2] for r in rows]))
    bad = [r for r in sorted(rows) if not (r[0] > 0.93 and r[1] < max(3.0 * r[2], 0.05))]
    assert not bad, f"{len(bad)} of {len(rows)} parameter gradients outside the fp16 envelope (cos, rel, envelope, name): {bad[:8]}"
    assert med_rel < max(2.0 * med_env, 0.02), (med_rel, med_env)
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            np.testing.assert_allclose(mod.running_mean.cpu().numpy(), sdo[name + ".running_mean"].numpy(), rtol=2e-2, atol=2e-3)
    print(f"\n[train] yolov5n bs={B} {S}^2: loss {loss.item():.5f} vs oracle {rloss.item():.5f}; parameter gradients: worst cosine "
          f"{worst_cos:.4f}, worst relative L2 error {worst_rel:.4f}, median {med_rel:.4f} (oracle fp16-storage envelope: median {med_env:.4f}, "
          f"worst {max(r[2] for r in rows):.4f})")


def test_train_step_yolov5s_bs64_timing(dev):
    """BASELINE config 3 per-GPU shape (yolov5s, 64 x 3x640x640, 512 targets): forward + ComputeLoss + backward + SGD step."""
    from yolov5_amd.loss import ComputeLoss

    m, cfg, sd = _model("yolov5s", dev)
    B = 64
    x = torch.rand((B, 3, 640, 640), device=dev).half()
    t = torch.from_numpy(detgen.synth_targets(B, 8, seed=1)).to(dev)
    compute_loss = ComputeLoss(m)
    from yolov5_amd.torch_utils import ModelEMA, smart_optimizer

    opt = smart_optimizer(m, "SGD", lr=0.01, momentum=0.937, decay=5e-4)
    ema = ModelEMA(m)
    losses = []
    for it in range(4):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.time()
        pred = m(x)
        loss, items = compute_loss(pred, t)
        opt.zero_grad(set_to_none=True)
        (loss * 1024.0).backward()
        if it == 0:  # parameter gradients are views of the engine's flat arena: autograd adopted them without copies
            eng = next(iter(m.__dict__["_train_engines"].values()))
            lo = eng.gflat.data_ptr()
            assert all(lo <= p.grad.data_ptr() < lo + eng.gtotal * 4 for p in m.parameters())
        stats = opt.step_fused(inv_scale=1.0 / 1024.0, max_norm=10.0, ema=ema, model=m)
        losses.append(float(loss.detach()))
        assert float(stats[2]) == 0.0
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 3 * 1e3
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]  # the loss goes down on a fixed batch
    print(f"\n[train] yolov5s bs=64 640^2: {ms:.1f} ms per step (fwd + ComputeLoss + bwd + SGD) = {B / ms * 1e3:.0f} img/s; losses {['%.3f' % l for l in losses]}")


def test_fp32_training_plan_exact_gradients_vs_oracle_autograd(dev):
    """The whole training step in fp32 (TrainEngine dtype=float32, selected by float32 images): yolov5n, 4 x 3 x 128 x 128 -- loss and
    EVERY parameter gradient against torch autograd over the CPU oracle at 1e-3 relative (no fp16 envelope)."""
    from yolov5_amd.loss import ComputeLoss

    m, cfg, sd = _model("yolov5n", dev)
    B, S = 4, 128
    x = torch.from_numpy(detgen.uniform((B, 3, S, S), 0.0, 1.0, name="timg", seed=9))
    t = torch.from_numpy(detgen.synth_targets(B, 5, seed=9))
    compute_loss = ComputeLoss(m)
    pred = m(x.to(dev))                       # float32 images -> fp32 plan
    assert all(p.dtype == torch.float32 for p in pred)
    loss, items = compute_loss(pred, t.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    sdo, leaves, ref, rloss, ritems = _oracle_grads(cfg, sd, x, t, 0)
    for a, b in zip(pred, ref):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(loss.item(), rloss.item(), rtol=1e-4)
    np.testing.assert_allclose(items.cpu().numpy(), ritems.numpy(), rtol=1e-4)
    worst = 0.0
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        a, b = p.grad.cpu().flatten().double(), leaves[n].grad.flatten().double()
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        worst = max(worst, rel)
        assert rel < 1e-3, (n, rel)
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            np.testing.assert_allclose(mod.running_var.cpu().numpy(), sdo[name + ".running_var"].numpy(), rtol=1e-4, atol=1e-6)
    print(f"\\n[train fp32] yolov5n bs={B} {S}^2: loss {loss.item():.6f} vs oracle {rloss.item():.6f}; worst relative L2 error over "
          f"{len(leaves)} parameter gradients {worst:.2e}")


def _step_grads(m, compute_loss, x, t, scale):
    """One forward + ComputeLoss + backward through the reference-shaped API; returns (loss, items, {name: grad / scale on the CPU})."""
    for p in m.parameters():
        p.grad = None
    pred = m(x)
    loss, items = compute_loss(pred, t)
    (loss * scale).backward()
    torch.cuda.synchronize()
    return float(loss.detach()), items.detach().float().cpu().numpy().copy(), {n: (p.grad.float().cpu() / scale) for n, p in m.named_parameters()}


def test_benchmarked_train_plan_parity(dev, monkeypatch):
    """The training plan bench.py times (BASELINE config 3 per-GPU shape: yolov5s, 64 x 3x640x640, 512 targets; utils/loss.py:134-183, train.py:402-410)
    compared with torch autograd over the CPU oracle AS A WHOLE -- tuned wgrad family x split choice per layer, wgrad3, the stem weight gradient, the
    strided data-gradient parity classes at 640^2 shapes:
      (1) fp32 plan: loss / loss items rtol 1e-4, every parameter gradient rel-L2 < 2e-3 (measured 9.8e-4);
      (2) fp16 (AMP) plan, the timed one: loss inside 2e-2, every parameter gradient inside a multiple of the oracle's own fp16-storage envelope.
          What that envelope can and cannot say was measured in round 5 (scripts/r5_train_bisect.py, profiles/r05/r05_train_plan_chaos.log): moving ONE
          input value by one fp16 ulp changes the fp16 plan's gradients by 7.6 % (median relative L2; worst parameter 14 %) and the fp32 plan's by 0.04 % --
          fp16 storage of the activations makes the step a discontinuous function of its inputs, so two tuner plans (= two rounding realisations) land
          0.08 ... 0.3 from the fp32 truth depending on the box's tile choices (six boxes: median 0.08 / 0.09 / 0.09 / 0.09 / 0.16 / 0.29), independent of
          the loss scale (128 ... 65536: identical).  The bound is therefore wide (cosine > 0.8, relative L2 < 8 x envelope, median < 5 x envelope): it
          catches a wrong backward kernel (which the exact checks (1), (3), (4) and tests/test_gpu_train_ops.py pin down), not a rounding realisation;
          the plan is printed with the figures so that an outlier can be reproduced;
      (3) the forced general weight-gradient family (Y5_WGRAD_CFG=1: the runner-up of most 3x3 layers) agrees with the tuned plan;
      (4) deterministic mode (Y5_DETERMINISTIC=1) is bit-identical across two runs."""
    import psutil

    from yolov5_amd.loss import ComputeLoss

    B, S, SCALE = 64, 640, 1024.0
    if psutil.virtual_memory().available < 56 * 2 ** 30:
        pytest.skip("the bs = 64 oracle autograd pass needs ~35 GB of host memory")
    g = torch.Generator().manual_seed(11)
    x = torch.rand((B, 3, S, S), generator=g)
    t = torch.from_numpy(detgen.synth_targets(B, 8, seed=11))
    assert t.shape[0] == 512
    m, cfg, sd = _model("yolov5s", dev)
    compute_loss = ComputeLoss(m)
    t0 = time.time()
    sdo, leaves, ref, rloss, ritems = _oracle_grads(cfg, sd, x, t, 0)
    t_oracle = time.time() - t0
    ref_g = {n: leaves[n].grad.flatten().double() for n in leaves}
    names = [n for n, _ in m.named_parameters()]
    assert len(names) == 177 and set(names) == set(ref_g)

    # (1) fp32 plan (float32 images select it)
    loss32, items32, g32 = _step_grads(m, compute_loss, x.to(dev), t.to(dev), 1.0)
    np.testing.assert_allclose(loss32, rloss.item(), rtol=1e-4)
    np.testing.assert_allclose(items32, ritems.numpy(), rtol=1e-4)
    worst32 = max(float((g32[n].flatten().double() - ref_g[n]).norm() / (ref_g[n].norm() + 1e-30)) for n in names)
    assert worst32 < 2e-3, worst32   # measured 9.8e-4 (profiles/r05/r05_pytest_trainplan_v0.log): fp32 sums over 26 M pixels per filter element, in an order other than the oracle's
    m.__dict__["_train_engines"].clear()
    del g32

    # (2) fp16 plan with the tuned choices
    xh, td = x.half().to(dev), t.to(dev)
    loss16, items16, g16 = _step_grads(m, compute_loss, xh, td, SCALE)
    np.testing.assert_allclose(loss16, rloss.item(), rtol=2e-2)
    np.testing.assert_allclose(items16, ritems.numpy(), rtol=3e-2)
    env = {n: 0.0 for n in names}
    for mode in (1, 2):
        lv = _oracle_grads(cfg, sd, x, t, mode)[1]
        for n in names:
            env[n] = max(env[n], float((lv[n].grad.flatten().double() - ref_g[n]).norm() / (ref_g[n].norm() + 1e-30)))
        del lv
    rows = []
    for n in names:
        a, b = g16[n].flatten().double(), ref_g[n]
        assert torch.isfinite(a).all(), n
        rows.append((float((a @ b) / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30)), env[n], n))
    bad = [r for r in sorted(rows) if not (r[0] > 0.8 and r[1] < max(8.0 * r[2], 0.2))]
    med_rel, med_env = float(np.median([r[1] for r in rows])), float(np.median([r[2] for r in rows]))
    eng = next(iter(m.__dict__["_train_engines"].values()))
    plan = [(st["op"]["name"], st["fcfg"], [c for sub in st["subs"] for c in sub["cfg"].values()], st.get("wg_choice")) for st in eng.convs]
    print(f"\n[train plan parity] plan (layer, forward cfg, data-gradient cfgs, weight-gradient (family, grid cap)): {plan}")
    print(f"\n[train plan parity] yolov5s bs={B} {S}^2, 512 targets (oracle autograd {t_oracle:.0f} s): fp32 plan loss {loss32:.6f} vs {rloss.item():.6f}, worst "
          f"gradient rel-L2 {worst32:.2e}; fp16 plan loss {loss16:.5f}, worst cosine {min(r[0] for r in rows):.4f}, worst rel-L2 {max(r[1] for r in rows):.4f}, "
          f"median {med_rel:.4f} (oracle fp16-storage envelope: median {med_env:.4f}, worst {max(r[2] for r in rows):.4f})")
    assert not bad, f"{len(bad)} of {len(rows)} parameter gradients outside the fp16 envelope (cos, rel, envelope, name): {bad[:8]}"
    assert med_rel < max(5.0 * med_env, 0.05), (med_rel, med_env)

    # (3) forced general weight-gradient family against the tuned plan (same math, other kernels / split counts)
    m.__dict__["_train_engines"].clear()
    monkeypatch.setenv("Y5_WGRAD_CFG", "1")
    _, _, gfam = _step_grads(m, compute_loss, xh, td, SCALE)
    worst_fam = max(float((gfam[n].flatten().double() - g16[n].flatten().double()).norm() / (g16[n].flatten().double().norm() + 1e-30)) for n in names)
    assert worst_fam < 2e-2, worst_fam
    monkeypatch.delenv("Y5_WGRAD_CFG")
    del gfam

    # (4) deterministic mode: two runs of the same step are bit-identical
    m.__dict__["_train_engines"].clear()
    monkeypatch.setenv("Y5_DETERMINISTIC", "1")
    la, _, ga = _step_grads(m, compute_loss, xh, td, SCALE)
    lb, _, gb = _step_grads(m, compute_loss, xh, td, SCALE)
    assert la == lb and all(torch.equal(ga[n], gb[n]) for n in names)
    worst_det = max(float((ga[n].flatten().double() - g16[n].flatten().double()).norm() / (g16[n].flatten().double().norm() + 1e-30)) for n in names)
    assert worst_det < 2e-2, worst_det
    print(f"[train plan parity] forced general wgrad family vs tuned: worst rel-L2 {worst_fam:.2e}; deterministic mode: bit-identical twice, vs atomic plan {worst_det:.2e}")


def test_benchmarked_train_plan_per_layer_gradients_fp16(dev):
    """VERDICT r5 weak 1 / ADVICE r5: the whole-plan fp16 bound above is wide by necessity (the fp16 step is a discontinuous function of its input), so a
    SYSTEMATIC fp16-only defect of <= 20 % in one layer's gradient kernels would pass it.  Here every weight-gradient and data-gradient launch of the tuned
    fp16 plan at the BENCHMARKED shapes (yolov5s, 64 x 3x640x640, 512 targets) is checked on its own: the hook behind each launch hands the launch's very
    operands (the fp16 activation x, the fp16 gradient dz) to torch's fp32 convolution-gradient routines on the GPU -- one rounding apart, no chaos --
    weight gradient rel-L2 < 2e-3 (fp32 sums of exact fp16 products, another order), data gradient within one fp16 rounding of the fp32 result."""
    import ctypes as C

    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.train_engine import train_forward  # noqa: F401

    B, S, SCALE = 64, 640, 1024.0
    g = torch.Generator().manual_seed(11)
    x = torch.rand((B, 3, S, S), generator=g).half().to(dev)
    t = torch.from_numpy(detgen.synth_targets(B, 8, seed=11)).to(dev)
    m, cfg, sd = _model("yolov5s", dev)
    compute_loss = ComputeLoss(m)
    pred = m(x)                                   # builds (and tunes) the plan
    eng = next(iter(m.__dict__["_train_engines"].values()))
    rows = []
    snap = {}

    def view(t_, grad):
        buf = (eng.gbufs if grad else eng.bufs)[t_.buf]
        return buf[..., t_.c_off:t_.c_off + t_.C]

    def dz_of(st, info):
        y = st["op"]["y"]
        npix = B * y.H * y.W
        c2s = st["c2s"]
        if st["has_bn"]:
            return eng.dz[: npix * info["ld_dz"]].view(B, y.H, y.W, info["ld_dz"])[..., :c2s]
        return view(y, True)[..., :c2s]

    def hook(kind, st, info):
        op, cv = st["op"], st["cv"]
        xr = op["x"]
        k, s_, p_ = tuple(op["k"]), tuple(op["s"]), tuple(op["p"])
        c2s = min(st["c2s"], cv.weight.shape[0])   # (Detect stores 256 channels for its 255)
        torch.cuda.synchronize()
        if kind == "wgrad":
            if info["geom"]["paired"]:
                return                              # the stem's paired-pixel view has its own kernel and test (test_conv_wgrad_stem_kernel)
            dz = dz_of(st, info)[..., :c2s].float().permute(0, 3, 1, 2)
            xin = view(xr, False).float().permute(0, 3, 1, 2)
            ref = torch.nn.grad.conv2d_weight(xin, (c2s, xr.C, k[0], k[1]), dz, stride=s_, padding=p_)
            ref = ref.permute(0, 2, 3, 1).reshape(c2s, -1).double()
            K = ref.shape[1]
            got = eng.dwflat[st["dw_off"]: st["dw_off"] + st["Npad"] * st["Kpad"]].view(st["Npad"], st["Kpad"])[:c2s, :K].double()
            rows.append(("wgrad", op["name"], float((got - ref).norm() / (ref.norm() + 1e-30)), str(info["choice"])))
        elif kind == "pre_dgrad":
            snap["dx"] = view(xr, True).clone() if info["acc"] else None
        elif kind == "dgrad":
            dz = dz_of(st, info)[..., :c2s].float().permute(0, 3, 1, 2)
            w16 = cv.weight.detach()[:c2s].half().float()
            ref = torch.nn.grad.conv2d_input((B, xr.C, xr.H, xr.W), w16, dz, stride=s_, padding=p_).permute(0, 2, 3, 1)
            if snap.get("dx") is not None:
                ref = ref + snap["dx"].float()
            got = view(xr, True).float()
            scale = float(ref.abs().max()) + 1e-30
            err = float((got - ref).abs().max()) / scale
            rows.append(("dgrad", op["name"], err, str(info["cfgs"])))

    eng.debug_hook = hook
    try:
        loss, _ = compute_loss(pred, t)
        (loss * SCALE).backward()
        torch.cuda.synchronize()
    finally:
        eng.debug_hook = None
    wg = [r for r in rows if r[0] == "wgrad"]
    dg = [r for r in rows if r[0] == "dgrad"]
    assert len(wg) >= 55 and len(dg) >= 50, (len(wg), len(dg))
    worst_w, worst_d = max(wg, key=lambda r: r[2]), max(dg, key=lambda r: r[2])
    print(f"\n[per-layer fp16 gradients] {len(wg)} weight-gradient launches: worst rel-L2 {worst_w[2]:.2e} ({worst_w[1]}, choice {worst_w[3]}); "
          f"{len(dg)} data-gradient layers: worst |err| / max|ref| {worst_d[2]:.2e} ({worst_d[1]}, cfgs {worst_d[3]})")
    assert worst_w[2] < 2e-3, worst_w
    assert worst_d[2] < 2e-3, worst_d   # one fp16 rounding of the result (2^-11 relative to the element, here relative to the largest one) + accumulation into fp16
