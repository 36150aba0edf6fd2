"""CPU: validation matching (yolov5_amd/csrc/metrics.hip, y5_val_match) on the HIP emulator, the oracle restatement and the
host-side ap_per_class against tests/golden/metrics.npz -- produced by the REFERENCE's own process_batch / scale_boxes /
ap_per_class (oracle/make_golden.py: gen_metrics)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import yolo_oracle as yo
from oracle.make_golden import METRIC_CASES, metrics_case
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import metrics as ym

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
IOUV = np.linspace(0.5, 0.95, 10).astype(np.float32)  # torch.linspace(0.5, 0.95, 10), val.py:222


def _bits(key, n):
    return np.unpackbits(G[key])[: n * 10].reshape(n, 10).astype(bool)


def _scale_rows(shapes):
    return np.array([[rp[0][0], rp[1][0], rp[1][1], s0[0], s0[1]] for s0, rp in shapes], np.float32)


def run_match(det, cnt, targets, scale, want_predn=False, label_layout="targets"):
    lib = emu()
    bs, max_det, ld = det.shape
    D = aligned(det.shape, np.float32); D[...] = det
    Cn = aligned((bs,), np.int32); Cn[...] = cnt
    T = aligned((max(targets.shape[0], 1), targets.shape[1]), np.float32); T[: targets.shape[0]] = targets
    I = aligned(IOUV.shape, np.float32); I[...] = IOUV
    S = None
    if scale is not None:
        S = aligned(scale.shape, np.float32); S[...] = scale
    out = aligned((bs, max_det, 10), np.uint8, 7)
    pn = aligned((bs, max_det, 4), np.float32, -1.0) if want_predn else None
    cols = (0, 1, 2, 1) if label_layout == "targets" else (-1, 0, 1, 0)
    rc = lib.y5_val_match(ptr(D), ld, max_det, ptr(Cn), bs, ptr(T) if targets.shape[0] else None, targets.shape[1], targets.shape[0], *cols,
                          ptr(S) if S is not None else None, ptr(I), 10, ptr(out), ptr(pn) if pn is not None else None, None)
    assert rc == 0, lib.y5_last_error()
    return out, pn


def test_iouv_is_torch_linspace():
    import torch
    assert np.array_equal(IOUV, torch.linspace(0.5, 0.95, 10).numpy())


@pytest.mark.parametrize("name", list(METRIC_CASES))
def test_oracle_process_batch_vs_reference_golden(name):
    det, cnt, t, shapes = metrics_case(name)
    for si in range(det.shape[0]):
        pred = det[si, : cnt[si]]
        lab = t[t[:, 0] == si, 1:]
        if lab.shape[0]:
            half = lab[:, 3:5] / np.float32(2)
            lab_xyxy = np.concatenate([lab[:, 0:1], lab[:, 1:3] - half, lab[:, 1:3] + half], 1)
            assert np.array_equal(yo.process_batch(pred, lab_xyxy, IOUV), _bits(f"{name}_{si}_direct", cnt[si]))
            correct, predn = yo.val_match_image(pred, lab, (640, 640), shapes[si][0], shapes[si][1], IOUV)
            assert np.array_equal(correct, _bits(f"{name}_{si}_correct", cnt[si]))
            assert np.array_equal(predn, G[f"{name}_{si}_predn"])


@pytest.mark.parametrize("name", list(METRIC_CASES))
def test_emu_val_match_vs_reference_golden(name):
    det, cnt, t, shapes = metrics_case(name)
    out, pn = run_match(det, cnt, t, _scale_rows(shapes), want_predn=True)
    direct, _ = run_match(det, cnt, t, None)
    for si in range(det.shape[0]):
        n = cnt[si]
        assert np.array_equal(out[si, :n].astype(bool), _bits(f"{name}_{si}_correct", n)), (name, si)
        assert np.array_equal(direct[si, :n].astype(bool), _bits(f"{name}_{si}_direct", n)), (name, si)
        assert np.array_equal(pn[si, :n], G[f"{name}_{si}_predn"])  # bit-exact de-letterbox
        assert not out[si, n:].any()  # rows past the count are cleared
        assert (pn[si, n:] == -1.0).all()  # and their predn rows untouched


def test_emu_process_batch_layout_and_tie_rule():
    """Single-image xyxy label layout (process_batch's own signature) and the tie rule: two identical labels, three identical
    detections -> the best label of each detection is the LATER label row, so only one detection (the first) is correct."""
    lab = np.array([[1, 10, 10, 50, 50], [1, 10, 10, 50, 50], [2, 100, 100, 150, 160]], np.float32)
    det = np.zeros((1, 8, 6), np.float32)
    det[0, 0] = [10, 10, 50, 50, 0.9, 1]
    det[0, 1] = [10, 10, 50, 50, 0.8, 1]
    det[0, 2] = [10, 10, 50, 50, 0.7, 1]
    det[0, 3] = [100, 100, 150, 158, 0.6, 2]  # IoU 0.9667 with label 2
    det[0, 4] = [100, 100, 150, 160, 0.5, 1]  # wrong class
    out, _ = run_match(det, np.array([5], np.int32), lab, None, label_layout="xyxy")
    ref = yo.process_batch(det[0, :5], lab, IOUV)
    assert np.array_equal(out[0, :5].astype(bool), ref)
    assert out[0, 0].all() and not out[0, 1].any() and not out[0, 2].any() and out[0, 3].all() and not out[0, 4].any()


def test_emu_val_match_many_labels_and_full_rows():
    """More labels than one LDS tile (256) spread over images in interleaved order, and max_det > 256 (several detections per lane)."""
    rng = np.random.default_rng(5)
    bs, max_det, M = 3, 600, 700
    t = np.zeros((M, 6), np.float32)
    t[:, 0] = rng.integers(0, bs, M)
    t[:, 1] = rng.integers(0, 3, M)
    t[:, 2:4] = rng.uniform(40, 600, (M, 2))
    t[:, 4:6] = rng.uniform(10, 90, (M, 2))
    det = np.zeros((bs, max_det, 6), np.float32)
    cnt = np.array([600, 431, 0], np.int32)
    for si in range(bs):
        lab = t[t[:, 0] == si]
        pick = rng.integers(0, lab.shape[0], max_det)
        c = lab[pick, 2:4] + rng.normal(0, 3, (max_det, 2))
        wh = lab[pick, 4:6] * rng.uniform(0.85, 1.15, (max_det, 2))
        det[si, :, 0:2] = c - wh / 2
        det[si, :, 2:4] = c + wh / 2
        det[si, :, 4] = np.sort(rng.uniform(0, 1, max_det))[::-1]
        det[si, :, 5] = lab[pick, 1]
    shapes = [((480, 640), ((0.8, 0.8), (0.0, 64.0)))] * bs
    out, _ = run_match(det, cnt, t, _scale_rows(shapes))
    for si in range(bs):
        n = cnt[si]
        ref, _ = yo.val_match_image(det[si, :n], t[t[:, 0] == si, 1:], (640, 640), shapes[si][0], shapes[si][1], IOUV)
        assert np.array_equal(out[si, :n].astype(bool), ref)
        assert not out[si, n:].any()


def test_val_match_rejects_bad_arguments():
    lib = emu()
    d = aligned((1, 4, 6), np.float32)
    o = aligned((1, 4, 10), np.uint8)
    I = aligned((10,), np.float32)
    assert lib.y5_val_match(ptr(d), 6, 2000, None, 1, None, 6, 0, 0, 1, 2, 1, None, ptr(I), 10, ptr(o), None, None) != 0  # max_det > 1024
    assert lib.y5_val_match(ptr(d), 6, 4, None, 1, None, 6, 0, 0, 1, 2, 1, None, ptr(I), 33, ptr(o), None, None) != 0  # niou > 32
    assert lib.y5_val_match(ptr(d), 6, 4, None, 1, None, 6, 3, 0, 1, 2, 1, None, ptr(I), 10, ptr(o), None, None) != 0  # labels NULL with nlabels > 0
    assert lib.y5_val_match(ptr(d), 6, 4, None, 1, ptr(d), 5, 1, 0, 1, 2, 1, None, ptr(I), 10, ptr(o), None, None) != 0  # box columns past the row


def _golden_stats():
    stats = []
    for name in METRIC_CASES:
        det, cnt, t, _ = metrics_case(name)
        for si in range(det.shape[0]):
            n = cnt[si]
            stats.append((_bits(f"{name}_{si}_correct", n), det[si, :n, 4], det[si, :n, 5], t[t[:, 0] == si, 1]))
    return [np.concatenate(x, 0) for x in zip(*stats)]


@pytest.mark.parametrize("impl", ["oracle", "host"])
def test_ap_per_class_vs_reference_golden(impl):
    tp, conf, pcls, tcls = _golden_stats()
    fn = yo.ap_per_class if impl == "oracle" else ym.ap_per_class
    res = fn(tp, conf, pcls, tcls)
    for k, v in zip(("tp", "fp", "p", "r", "f1", "ap", "cls"), res):
        np.testing.assert_allclose(np.asarray(v, np.float64), G["ap_" + k].astype(np.float64), rtol=0, atol=1e-12, err_msg=k)


def test_host_smooth_compute_ap_fitness():
    from oracle.thirdparty import smooth as osmooth
    y = np.sin(np.linspace(0, 7, 1000)) ** 2
    np.testing.assert_allclose(ym.smooth(y, 0.1), osmooth(y, 0.1), rtol=0, atol=1e-15)
    rec = np.linspace(0.01, 0.9, 57)
    pre = np.cos(rec) * 0.9
    assert abs(ym.compute_ap(rec, pre)[0] - yo.compute_ap(rec, pre)[0]) < 1e-15
    assert np.allclose(ym.fitness(np.array([[0.5, 0.6, 0.7, 0.8, 9.0]])), [0.79])


def test_emu_scale_boxes_batch_vs_reference_golden():
    """tests/golden/scale_boxes.npz: the reference's scale_boxes on 20 boxes, (640,640)->(1080,810) and (384,640)->(720,1280)."""
    from oracle import detgen

    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "scale_boxes.npz"))
    b = detgen.uniform((20, 4), -20, 660, name="sb", seed=14)
    lib = emu()
    det = aligned((2, 32, 7), np.float32, 5.0)
    det[:, :20, :4] = b
    cnt = aligned((2,), np.int32); cnt[...] = [20, 20]
    rows = []
    for img1, img0 in (((640, 640), (1080, 810)), ((384, 640), (720, 1280))):
        gain = min(img1[0] / img0[0], img1[1] / img0[1])
        rows.append([gain, (img1[1] - img0[1] * gain) / 2, (img1[0] - img0[0] * gain) / 2, img0[0], img0[1]])
    sc = aligned((2, 5), np.float32); sc[...] = np.array(rows, np.float32)
    rnd = det.copy()
    assert lib.y5_scale_boxes_batch(ptr(det), 7, 32, ptr(cnt), 2, ptr(sc), 0, None) == 0, lib.y5_last_error()
    assert np.array_equal(det[0, :20, :4], S["a"]) and np.array_equal(det[1, :20, :4], S["b"])
    assert (det[:, 20:] == 5.0).all() and (det[:, :, 4:] == 5.0).all()  # rows past the count and other columns untouched
    rnd_buf = aligned(rnd.shape, np.float32); rnd_buf[...] = rnd
    assert lib.y5_scale_boxes_batch(ptr(rnd_buf), 7, 32, ptr(cnt), 2, ptr(sc), 1, None) == 0
    import torch
    assert np.array_equal(rnd_buf[0, :20, :4], torch.from_numpy(S["a"]).round().numpy())  # detect.py:248
    assert lib.y5_scale_boxes_batch(None, 7, 32, ptr(cnt), 2, ptr(sc), 0, None) != 0
