"""CPU, this container only (skipped where /root/reference is absent, e.g. on the GPU box): the oracle restatement
(oracle/yolo_oracle.py) against the UNMODIFIED reference modules imported through oracle/ref_shim.py, on inputs that
differ from the committed golden fixtures."""
import numpy as np
import pytest
import torch

from oracle import detgen, ref_shim, yolo_oracle as yo

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ns():
    yield ref_shim.load()
    ref_shim.unload()   # the next test file in this xdist worker starts without the reference's `models` / `utils` packages in sys.modules


def test_build_targets_and_loss_match_live_reference(ns):
    import os

    import yaml

    torch.manual_seed(0)
    m = ns.yolo.DetectionModel(os.path.join(ns.root, "models/yolov5s.yaml"))
    with open(os.path.join(ns.root, "data/hyps/hyp.scratch-low.yaml")) as f:
        m.hyp = yaml.safe_load(f)
    cl = ns.loss.ComputeLoss(m)
    anchors = m.model[-1].anchors.clone()
    pn = [detgen.uniform((3, 3, s, s, 85), -3.0, 3.0, name=f"v{s}", seed=31) for s in (24, 12, 6)]
    tn = detgen.synth_targets(3, 11, seed=31)
    p1 = [torch.from_numpy(a).clone().requires_grad_(True) for a in pn]
    p2 = [torch.from_numpy(a).clone().requires_grad_(True) for a in pn]
    t = torch.from_numpy(tn)
    r_tcls, r_tbox, r_idx, r_anch = cl.build_targets(p1, t)
    o_tcls, o_tbox, o_idx, o_anch = yo.build_targets([q.shape for q in p2], t, anchors)
    for i in range(3):
        assert all(torch.equal(a, b) for a, b in zip(r_idx[i], o_idx[i]))
        assert torch.equal(r_tcls[i], o_tcls[i]) and torch.equal(r_tbox[i], o_tbox[i]) and torch.equal(r_anch[i], o_anch[i])
    rl, ri = cl(p1, t)
    ol, oi = yo.compute_loss(p2, t, anchors)
    rl.backward()
    ol.backward()
    np.testing.assert_allclose(ol.detach().numpy(), rl.detach().numpy(), rtol=1e-6)
    np.testing.assert_allclose(oi.numpy(), ri.numpy(), rtol=1e-6)
    for a, b in zip(p1, p2):
        np.testing.assert_allclose(b.grad.numpy(), a.grad.numpy(), rtol=1e-5, atol=1e-10)


def test_autobalance_matches_live_reference(ns):
    """ComputeLoss(autobalance=True) (utils/loss.py:127, :173-177): four calls of the live reference against the oracle's restatement -- losses and
    the drifting balance list (Python doubles) step by step."""
    import os

    import yaml

    m = ns.yolo.DetectionModel(os.path.join(ns.root, "models/yolov5s.yaml"), ch=3, nc=80)
    with open(os.path.join(ns.root, "data/hyps/hyp.scratch-low.yaml")) as f:
        m.hyp = yaml.safe_load(f)
    cl = ns.loss.ComputeLoss(m, autobalance=True)
    assert cl.ssi == 1
    anchors = m.model[-1].anchors.clone()
    bal = [4.0, 1.0, 0.4]
    for step in range(4):
        p = [torch.from_numpy(detgen.uniform((2, 3, s, s, 85), -3.0, 3.0, name=f"ab{s}", seed=40 + step)) for s in (16, 8, 4)]
        t = torch.from_numpy(detgen.synth_targets(2, 6, seed=40 + step))
        loss_r, items_r = cl(p, t)
        loss_o, items_o = yo.compute_loss(p, t, anchors, nc=80, balance=bal, autobalance_ssi=1)
        torch.testing.assert_close(loss_o, loss_r, rtol=1e-6, atol=0)
        torch.testing.assert_close(items_o, items_r, rtol=1e-6, atol=0)
        assert bal == cl.balance, (step, bal, cl.balance)
    assert bal[1] == 1.0 and bal[0] != 4.0


def test_forward_augment_matches_live_reference(ns):
    """models/yolo.py:269-312 (TTA: scales 1 / 0.83 / 0.67 + left-right flip, _descale_pred, _clip_augmented): the live reference's
    model(x, augment=True) against oracle.forward_augment on the same weights."""
    import os

    torch.manual_seed(0)
    m = ns.yolo.DetectionModel(os.path.join(ns.root, "models/yolov5n.yaml"))
    cfg = yo.model_cfg("yolov5n")
    sd = yo.det_state_dict(cfg, 3, fused=False)
    m.load_state_dict(sd)
    m.eval()
    x = torch.from_numpy(detgen.uniform((2, 3, 96, 128), 0.0, 1.0, name="tta", seed=5))
    with torch.no_grad():
        zr = m(x, augment=True)[0]
        zo = yo.forward_augment(cfg, sd, x)
    assert zr.shape == zo.shape
    torch.testing.assert_close(zo, zr, rtol=1e-5, atol=1e-5)


def test_nms_matches_live_reference(ns):
    pred = detgen.synth_predictions(2, 1500, 85, obj_pow=3, seed=41)
    with ref_shim.oracle_nms_mode():
        ref = ns.general.non_max_suppression(torch.from_numpy(pred), 0.25, 0.45, max_det=300)
    out = yo.non_max_suppression(pred, 0.25, 0.45, max_det=300)
    for r, o in zip(ref, out):
        assert np.array_equal(r.numpy(), o)


def test_nms_with_apriori_labels_matches_live_reference(ns):
    """general.py:706-712 (`labels`, val.py --save-hybrid): the a-priori label rows join the candidates with confidence 1; one image without labels,
    multi_label on and off."""
    pred = detgen.synth_predictions(3, 800, 15, obj_pow=3, seed=43)
    labels = [np.array([[2, 100.0, 120.0, 40.0, 60.0], [7, 300.5, 310.25, 80.0, 20.0]], np.float32), np.zeros((0, 5), np.float32),
              np.array([[0, 50.0, 60.0, 30.0, 30.0]], np.float32)]
    for ml in (False, True):
        with ref_shim.oracle_nms_mode():
            ref = ns.general.non_max_suppression(torch.from_numpy(pred), 0.25, 0.45, labels=[torch.from_numpy(lb) for lb in labels], multi_label=ml, max_det=300)
        out = yo.non_max_suppression(pred, 0.25, 0.45, labels=labels, multi_label=ml, max_det=300)
        for r, o in zip(ref, out):
            assert np.array_equal(r.numpy(), o)
        assert (out[0][:2, 4] == 1.0).all() and {int(c) for c in out[0][:2, 5]} == {2, 7}


def test_process_batch_and_ap_match_live_reference(ns):
    """utils/metrics.py process_batch / ap_per_class (unmodified) vs the oracle restatement on a batch that is not in the fixtures."""
    rng = np.random.default_rng(77)
    iouv = torch.linspace(0.5, 0.95, 10)
    stats = []
    for _ in range(4):
        m = int(rng.integers(1, 25))
        lab = np.zeros((m, 5), np.float32)
        lab[:, 0] = rng.integers(0, 4, m)
        xy = rng.uniform(0, 500, (m, 2)); wh = rng.uniform(10, 140, (m, 2))
        lab[:, 1:3], lab[:, 3:5] = xy, xy + wh
        pick = rng.integers(0, m, 90)
        det = np.zeros((90, 6), np.float32)
        det[:, :4] = lab[pick, 1:] + rng.normal(0, 5, (90, 4))
        det[:, 4] = np.sort(rng.uniform(0, 1, 90))[::-1]
        det[:, 5] = np.where(rng.uniform(0, 1, 90) < 0.8, lab[pick, 0], rng.integers(0, 4, 90))
        ref = ns.metrics.process_batch(torch.from_numpy(det), torch.from_numpy(lab), iouv).numpy()
        assert np.array_equal(yo.process_batch(det, lab, iouv.numpy()), ref)
        stats.append((ref, det[:, 4], det[:, 5], lab[:, 0]))
    tp, conf, pcls, tcls = (np.concatenate(x, 0) for x in zip(*stats))
    r = ns.metrics.ap_per_class(tp, conf, pcls, tcls, plot=False, names={})
    o = yo.ap_per_class(tp, conf, pcls, tcls)
    for a, b in zip(r, o):
        np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=0, atol=1e-12)


def test_training_loop_restatement_matches_reference_pieces(ns):
    """oracle/train_oracle.py (the fp32 yardstick of the train-loop tests) against the same schedule driven through the REFERENCE's own
    DetectionModel (train mode), ComputeLoss, smart_optimizer and ModelEMA (train.py:234-248,372-434): per-iteration loss items,
    final parameters and EMA must agree to fp32 round-off."""
    import copy

    from oracle import train_oracle as to
    from oracle.make_golden import TINY_CFG, load_det_weights

    imgs, tpi = to.synthetic_set(8, 64, per_img=2, seed=6)
    bs, epochs = 4, 3
    hyp = dict(to.HYP)
    cfg = copy.deepcopy(TINY_CFG)
    sd = yo.det_state_dict(cfg, 9, fused=False)
    ref = to.train_oracle(cfg, sd, imgs, tpi, bs, hyp=dict(hyp), epochs=epochs, cos_lr=True)

    torch.manual_seed(0)
    m = ns.yolo.DetectionModel(copy.deepcopy(TINY_CFG))
    load_det_weights(m, 9)
    h = dict(hyp)
    nbs, nb = 64, 2
    accumulate = max(round(nbs / bs), 1)
    h["weight_decay"] *= bs * accumulate / nbs
    opt = ns.torch_utils.smart_optimizer(m, "SGD", h["lr0"], h["momentum"], h["weight_decay"])
    import math

    lf = lambda x: ((1 - math.cos(x * math.pi / epochs)) / 2) * (h["lrf"] - 1) + 1  # noqa: E731  one_cycle(1, lrf, epochs)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lf)
    ema = ns.torch_utils.ModelEMA(m)
    m.hyp = h
    cl = ns.loss.ComputeLoss(m)
    nw = max(round(h["warmup_epochs"] * nb), 100)
    last, losses = -1, []
    for epoch in range(epochs):
        m.train()
        opt.zero_grad()
        for i in range(nb):
            ids = list(range(i * bs, (i + 1) * bs))
            ni = i + nb * epoch
            x = imgs[ids].float() / 255
            t = torch.cat([torch.cat((torch.full((len(tpi[j]), 1), float(k)), tpi[j][:, 1:]), 1) for k, j in enumerate(ids)], 0)
            if ni <= nw:
                accumulate = max(1, np.interp(ni, [0, nw], [1, nbs / bs]).round())
                for j, g in enumerate(opt.param_groups):
                    g["lr"] = np.interp(ni, [0, nw], [h["warmup_bias_lr"] if j == 0 else 0.0, g["initial_lr"] * lf(epoch)])
                    g["momentum"] = np.interp(ni, [0, nw], [h["warmup_momentum"], h["momentum"]])
            loss, items = cl(m(x), t)
            loss.backward()
            if ni - last >= accumulate:
                torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=10.0)
                opt.step()
                opt.zero_grad()
                ema.update(m)
                last = ni
            losses.append(items.detach().clone())
        sched.step()
    np.testing.assert_allclose(ref["losses"].numpy(), torch.stack(losses).numpy(), rtol=2e-4, atol=1e-6)
    assert ref["updates"] == ema.updates
    msd, esd = m.state_dict(), ema.ema.state_dict()
    for k, v in ref["sd"].items():
        if v.dtype.is_floating_point and not k.endswith("anchors"):
            np.testing.assert_allclose(v.numpy(), msd[k].numpy(), rtol=2e-3, atol=2e-5, err_msg=k)
            np.testing.assert_allclose(ref["ema"][k].numpy(), esd[k].numpy(), rtol=2e-3, atol=2e-5, err_msg="ema " + k)
