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
    return ref_shim.load()


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


def test_nms_matches_live_reference(ns):
    pred = detgen.synth_predictions(2, 1500, 85, obj_pow=3, seed=41)
    with ref_shim.oracle_nms_mode():
        ref = ns.general.non_max_suppression(torch.from_numpy(pred), 0.25, 0.45, max_det=300)
    out = yo.non_max_suppression(pred, 0.25, 0.45, max_det=300)
    for r, o in zip(ref, out):
        assert np.array_equal(r.numpy(), o)
