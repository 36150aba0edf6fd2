"""GPU parity of the validation matching (HIP y5_val_match through yolov5_amd.metrics) vs the reference-generated golden
fixture (bit-exact) and, end to end behind non_max_suppression, vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo
from oracle.make_golden import METRIC_CASES, metrics_case

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _bits(key, n):
    return np.unpackbits(G[key])[: n * 10].reshape(n, 10).astype(bool)


@pytest.mark.parametrize("name", list(METRIC_CASES))
def test_match_batch_vs_reference_golden(name, dev):
    from yolov5_amd.metrics import match_batch, process_batch

    det, cnt, t, shapes = metrics_case(name)
    iouv = torch.linspace(0.5, 0.95, 10, device=dev)
    out, counts, targets = torch.from_numpy(det).to(dev), torch.from_numpy(cnt).to(dev), torch.from_numpy(t).to(dev)
    correct, predn = match_batch(out, counts, targets, shapes, iouv, predn=True)
    direct = match_batch(out, counts, targets, None, iouv)
    correct, predn, direct = correct.cpu().numpy(), predn.cpu().numpy(), direct.cpu().numpy()
    for si in range(det.shape[0]):
        n = cnt[si]
        assert np.array_equal(correct[si, :n].astype(bool), _bits(f"{name}_{si}_correct", n)), (name, si)
        assert np.array_equal(direct[si, :n].astype(bool), _bits(f"{name}_{si}_direct", n)), (name, si)
        assert np.array_equal(predn[si, :n], G[f"{name}_{si}_predn"])
        assert not correct[si, n:].any()
        # the reference-signature entry point on one image
        lab = targets[targets[:, 0] == si, 1:]
        half = lab[:, 3:5] / 2
        lab_xyxy = torch.cat((lab[:, 0:1], lab[:, 1:3] - half, lab[:, 1:3] + half), 1)
        pb = process_batch(out[si, :n], lab_xyxy, iouv)
        assert pb.dtype == torch.bool and np.array_equal(pb.cpu().numpy(), _bits(f"{name}_{si}_direct", n))


def test_val_stats_behind_nms_vs_oracle(dev):
    """val.py:276-309 on the device: NMS (padded, no host sync) -> ValStats.update -> compute, against the oracle's NMS +
    per-image matching + ap_per_class on the same synthetic predictions."""
    from yolov5_amd.general import non_max_suppression
    from yolov5_amd.metrics import ValStats

    bs, nc = 4, 6
    pred = detgen.synth_predictions(bs, 4000, 5 + nc, obj_pow=6, seed=41)
    t = detgen.synth_targets(bs, 12, nc, seed=41)
    t[:, 2:] *= np.float32(640)
    # plant rows on the labels so that there are true positives
    for si in range(bs):
        lab = t[t[:, 0] == si]
        for l in range(lab.shape[0]):
            for k in range(3):
                r = si * 0 + l * 3 + k
                pred[si, r, :4] = lab[l, 2:6] + np.float32(k)
                pred[si, r, 4] = 0.9 - 0.1 * k
                pred[si, r, 5:] = 0.02
                pred[si, r, 5 + int(lab[l, 1])] = 0.95
    shapes = [((480, 640), ((0.8, 0.8), (0.0, 64.0))), ((640, 640), ((1.0, 1.0), (0.0, 0.0))),
              ((300, 400), ((1.6, 1.6), (0.0, 80.0))), ((1080, 810), ((0.5925926, 0.5925926), (80.0, 0.0)))]
    iouv = torch.linspace(0.5, 0.95, 10, device=dev)
    p = torch.from_numpy(pred).to(dev)
    out, counts = non_max_suppression(p, 0.001, 0.6, multi_label=True, max_det=300, padded=True)
    vs = ValStats(iouv)
    vs.update(out, counts, torch.from_numpy(t).to(dev), shapes)
    res = vs.compute(nc=nc)
    ref_out = yo.non_max_suppression(pred, 0.001, 0.6, multi_label=True, max_det=300)
    stats = []
    for si in range(bs):
        lab = t[t[:, 0] == si, 1:]
        c, _ = yo.val_match_image(ref_out[si], lab, (640, 640), shapes[si][0], shapes[si][1], iouv.cpu().numpy())
        stats.append((c, ref_out[si][:, 4], ref_out[si][:, 5], lab[:, 0]))
    tp, conf, pcls, tcls = (np.concatenate(x, 0) for x in zip(*stats))
    assert np.array_equal(torch.cat(vs.correct).cpu().numpy(), tp)
    _, _, pp, rr, _, ap, cls = yo.ap_per_class(tp, conf, pcls, tcls)
    assert tp[:, 0].sum() >= 40
    np.testing.assert_allclose(res["ap"], ap, rtol=0, atol=1e-12)
    assert abs(res["map"] - ap.mean()) < 1e-12 and abs(res["mp"] - pp.mean()) < 1e-12 and abs(res["mr"] - rr.mean()) < 1e-12
    assert np.array_equal(res["nt"], np.bincount(tcls.astype(int), minlength=nc))


def test_metrics_need_gpu():
    from yolov5_amd.metrics import match_batch, process_batch

    with pytest.raises(RuntimeError):
        process_batch(torch.zeros(3, 6), torch.zeros(2, 5), torch.linspace(0.5, 0.95, 10))
    with pytest.raises(RuntimeError):
        match_batch(torch.zeros(1, 3, 6), torch.zeros(1, dtype=torch.int32), torch.zeros(0, 6), None, torch.linspace(0.5, 0.95, 10))


def test_scale_boxes_batch_vs_reference_golden(dev):
    from yolov5_amd.general import scale_boxes_batch

    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "scale_boxes.npz"))
    b = detgen.uniform((20, 4), -20, 660, name="sb", seed=14)
    out = torch.full((1, 32, 6), 5.0, device=dev)
    out[0, :20, :4] = torch.from_numpy(b).to(dev)
    cnt = torch.tensor([20], dtype=torch.int32, device=dev)
    a = scale_boxes_batch((640, 640), out.clone(), cnt, [(1080, 810)])
    assert np.array_equal(a[0, :20, :4].cpu().numpy(), S["a"]) and bool((a[0, 20:] == 5.0).all()) and bool((a[0, :, 4:] == 5.0).all())
    bb = scale_boxes_batch((384, 640), out.clone(), cnt, [(720, 1280)], round_=True)
    assert np.array_equal(bb[0, :20, :4].cpu().numpy(), torch.from_numpy(S["b"]).round().numpy())
    # explicit ratio_pad (val.py:298)
    c = scale_boxes_batch((640, 640), out.clone(), cnt, [(1080, 810)], ratio_pads=[((640 / 1080, 640 / 1080), (80.0, 0.0))])
    assert np.array_equal(c[0, :20, :4].cpu().numpy(), S["a"])
