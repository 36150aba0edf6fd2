"""GPU (-m gpu): the reference-shaped loops on the MI355X.
  * C3 in miniature: yolov5n trained for 50 iterations on a fixed 32-image synthetic set by yolov5_amd.train_loop.train (warm-up,
    accumulate, LambdaLR, loss scaling, fused clip + SGD + EMA) against oracle/train_oracle.py (fp32 torch-CPU autograd running the
    same schedule, pinned to the reference's pieces in tests/test_oracle_vs_reference.py): schedule identical, loss curve within the
    fp16 band, the model actually learns (loss falls), EMA / BatchNorm statistics track the oracle's;
  * detect(): letterbox -> forward -> NMS -> scale_boxes for a batch of frames vs the oracle pipeline (detect.py:204-248);
  * DetectPipeline: the deferred-collect pipeline returns exactly what the sequential calls return."""
import numpy as np
import pytest
import torch

from oracle import detgen, train_oracle as to, yolo_oracle as yo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_train_loop_50_iterations_vs_oracle(dev):
    from yolov5_amd.train_loop import TensorLoader, train
    from yolov5_amd.yolo import DetectionModel

    cfg = yo.model_cfg("yolov5n")
    sd = yo.det_state_dict(cfg, 0, fused=False)
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(sd)
    m = m.to(dev)
    imgs, tpi = to.synthetic_set(32, 128, per_img=3, seed=1)
    bs, epochs = 8, 13                                   # 4 batches per epoch -> 52 iterations
    hyp = dict(to.HYP)
    res = train(m, TensorLoader(imgs.to(dev), [t.to(dev) for t in tpi], bs), hyp=dict(hyp), epochs=epochs, device=dev, amp=True)
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = to.train_oracle(cfg, sd, imgs, tpi, bs, hyp=dict(hyp), epochs=epochs)
    a, b = res["losses"].numpy(), ref["losses"].numpy()
    assert a.shape == b.shape == (52, 3)
    np.testing.assert_allclose(np.array(res["lr"]), np.array(ref["lr"]), rtol=1e-12)        # same schedule
    assert res["ema"].updates == ref["updates"]
    rel = np.abs(a.sum(1) - b.sum(1)) / b.sum(1)
    print(f"\\n[train-loop] total loss first/last: hip {a.sum(1)[0]:.4f}/{a.sum(1)[-4:].mean():.4f}  oracle {b.sum(1)[0]:.4f}/{b.sum(1)[-4:].mean():.4f}; "
          f"rel. deviation max {rel.max():.4f} mean {rel.mean():.4f}; scaler scale {res['scaler'].scale}, skipped {res['scaler'].skipped}")
    assert rel[:8].max() < 0.03 and rel.mean() < 0.03 and rel.max() < 0.10, (rel.max(), rel.mean())
    # it trains: the last epoch's mean loss is clearly below the first epoch's, for both
    assert a.sum(1)[-4:].mean() < 0.9 * a.sum(1)[:4].mean() and b.sum(1)[-4:].mean() < 0.9 * b.sum(1)[:4].mean()
    assert abs(a.sum(1)[-4:].mean() - b.sum(1)[-4:].mean()) < 0.03 * b.sum(1)[-4:].mean()     # final loss within 3 %
    # EMA and BatchNorm running statistics follow the oracle's
    esd = res["ema"].ema.state_dict()
    errs = []
    for k, v in ref["ema"].items():
        if v.dtype.is_floating_point and not k.endswith("anchors"):
            errs.append((float((esd[k].float().cpu() - v).norm() / (v.norm() + 1e-12)), k))
    errs.sort(reverse=True)
    print("[train-loop] EMA relative L2 deviation, worst tensors:", [(round(e, 4), k) for e, k in errs[:5]], "median", round(errs[len(errs) // 2][0], 5))
    # (the small early-layer BatchNorm biases move with the 0.1 warm-up bias lr on fp16-noisy gradients -- DESIGN.md section 4,
    # "ill-conditioned at fp16 resolution"; the loss curve above is the functional criterion, this one bounds the drift)
    assert errs[len(errs) // 2][0] < 0.02 and errs[0][0] < 0.6, errs[:5]


def test_train_loop_fp32_mode_tracks_oracle_per_step(dev):
    """train(..., amp=False): the fp32 training plan under the same schedule -- every one of the 24 per-iteration loss triples within 1e-3
    of the oracle loop's (oracle/train_oracle.py), parameters after training within 1e-3 relative."""
    from yolov5_amd.train_loop import TensorLoader, train
    from yolov5_amd.yolo import DetectionModel

    cfg = yo.model_cfg("yolov5n")
    sd = yo.det_state_dict(cfg, 1, fused=False)
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(sd)
    m = m.to(dev)
    imgs, tpi = to.synthetic_set(16, 128, per_img=3, seed=2)
    bs, epochs = 4, 6
    hyp = dict(to.HYP)
    res = train(m, TensorLoader(imgs.to(dev), [t.to(dev) for t in tpi], bs), hyp=dict(hyp), epochs=epochs, device=dev, amp=False)
    ref = to.train_oracle(cfg, sd, imgs, tpi, bs, hyp=dict(hyp), epochs=epochs)
    a, b = res["losses"].numpy(), ref["losses"].numpy()
    assert a.shape == b.shape == (24, 3)
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
    print(f"\n[train-loop fp32] per-step loss deviation from the oracle: max {rel.max():.2e}, mean {rel.mean():.2e}; final loss {a.sum(1)[-1]:.5f} vs {b.sum(1)[-1]:.5f}")
    assert rel.max() < 1e-3, rel.max()
    worst = max(float((p.detach().cpu() - ref["sd"][k]).norm() / (ref["sd"][k].norm() + 1e-12)) for k, p in m.named_parameters())
    assert worst < 1e-3, worst


def test_detect_loop_batch_vs_oracle_pipeline(dev):
    from yolov5_amd.detect_loop import detect
    from yolov5_amd.yolo import DetectionModel

    cfg = yo.model_cfg("yolov5n")
    sd = yo.det_state_dict(cfg, 7, fused=False)
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(sd)
    det = m.model[-1]
    with torch.no_grad():
        for mi in det.m:
            b = mi.bias.view(det.na, -1)
            b[:, 4] += 1.5
            b[:, 5:] += 1.0
    m = m.eval().fuse().to(dev)
    sd_f = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(3)
    ims = [rng.integers(0, 256, s, dtype=np.uint8) for s in ((240, 320, 3), (300, 200, 3), (128, 128, 3), (90, 400, 3))]
    out = detect(m, ims, imgsz=320, conf_thres=0.25, iou_thres=0.45, max_det=300)
    assert len(out) == 4
    for im, got in zip(ims, out):
        lb, _, _ = yo.letterbox(im, (320, 320), auto=False)
        x = torch.from_numpy(np.ascontiguousarray(lb.transpose(2, 0, 1))[None]).float() / 255
        with torch.no_grad():
            z = yo.model_forward(cfg, sd_f, x)[0]
        e = yo.non_max_suppression(z.numpy(), 0.25, 0.45, max_det=300)[0].copy()
        yo.scale_boxes((320, 320), e[:, :4], im.shape[:2])
        e[:, :4] = np.round(e[:, :4])
        from tests import detset

        assert abs(len(got) - len(e)) <= max(1, 0.02 * len(e)) and len(e) > 5, (len(got), len(e))
        # same detections (rows of nearly equal confidence may come in a different order: fp32 round-off between two forwards)
        a = detset.agreement(e, got.numpy(), 0.25, box_atol=1.0, conf_atol=1e-3, margin=1e-3)
        assert a["unmatched_ref"] + a["unmatched_got"] <= 0.02 * (a["ref_strong"] + a["got_strong"]) + 1, a


def test_detect_pipeline_equals_sequential(dev):
    from yolov5_amd.detect_loop import DetectPipeline
    from yolov5_amd.general import non_max_suppression
    from yolov5_amd.yolo import DetectionModel

    torch.manual_seed(0)
    m = DetectionModel("yolov5n.yaml")
    det = m.model[-1]
    with torch.no_grad():
        for mi in det.m:
            b = mi.bias.view(det.na, -1)
            b[:, 4] += 6.0
            b[:, 5:] += 6.0
    m = m.eval().fuse().half().to(dev)
    det.export = True
    xs = [torch.rand((4, 3, 256, 256), device=dev).half() for _ in range(5)]
    seq = [non_max_suppression(m(x)[0].clone(), 0.25, 0.45, max_det=300) for x in xs]
    for overlap in (True, False):   # NMS on the high-priority side stream beside the next forward / everything on one stream
        pipe = DetectPipeline(m, 0.25, 0.45, max_det=300, overlap=overlap)
        got = []
        for x in xs + xs:
            r = pipe.submit(x)
            if r is not None:
                got.append(r)
        got.append(pipe.flush())
        assert len(got) == 2 * len(seq) == 10
        for a, b in zip(seq + seq, got):
            assert len(a) == len(b) == 4
            for u, v in zip(a, b):
                assert torch.equal(u, v)
    assert sum(len(u) for a in seq for u in a) > 20


def _inside_picks(d, n=3, S=64):
    """Three of the oracle's own detections to serve as labels: boxes whose CENTRE lies inside the image.  (With the objectness saturated the top
    confidences tie to the last bit, so 'rows 0, 5, 11' named different -- sometimes off-image -- boxes on different host CPUs.)"""
    cx, cy = (d[:, 0] + d[:, 2]) / 2, (d[:, 1] + d[:, 3]) / 2
    ok = (cx > 4) & (cx < S - 4) & (cy > 4) & (cy < S - 4)
    idx = np.flatnonzero(ok)
    idx = idx[np.linspace(0, len(idx) - 1, n).astype(int)] if len(idx) >= n else np.arange(n)
    return d[idx]


def test_val_loop_on_gpu_vs_oracle_pipeline(dev):
    """val.py:255-333 through yolov5_amd.val_loop.run on the MI355X (fp32 model so that the comparison is exact in the matching; the fp16 forward
    is covered by the detection-set tests): P / R / mAP@.5 / mAP@.5:.95 and the validation loss against the oracle's forward -> NMS(conf 0.001,
    iou 0.6, multi_label) -> per-image scale_boxes + process_batch -> ap_per_class on the same images, labels and letterbox geometry."""
    from tests.test_loops import _ValLoader, _val_set  # noqa: F401
    from yolov5_amd import val_loop
    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.yolo import DetectionModel

    cfg = yo.model_cfg("yolov5n")
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(yo.det_state_dict(cfg, 0, fused=False))
    m.hyp = dict(to.HYP)
    with torch.no_grad():
        for mi in m.model[-1].m:
            mi.bias.view(m.model[-1].na, -1)[:, 4:] += 3.0
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    imgs, tpi, shapes = _val_set(8, 64, seed=9)
    nc = m.model[-1].nc
    with torch.no_grad():
        z0 = yo.model_forward(cfg, sd, imgs.float() / 255)[0].numpy()
    d0 = yo.non_max_suppression(z0, 0.001, 0.6, multi_label=True, max_det=300)
    for i, d in enumerate(d0):
        pick = _inside_picks(d)
        xywh = np.stack([(pick[:, 0] + pick[:, 2]) / 2 + 1.0 + i % 2, (pick[:, 1] + pick[:, 3]) / 2 - 1.0, (pick[:, 2] - pick[:, 0]) * 1.05, pick[:, 3] - pick[:, 1]], 1) / 64.0
        tpi[i] = torch.from_numpy(np.concatenate([np.zeros((3, 1), np.float32), pick[:, 5:6], xywh.astype(np.float32)], 1))
    m = m.to(dev)
    loader = _ValLoader(imgs, tpi, shapes, 4)
    (mp, mr, map50, map_, lb, lo, lc), maps, t = val_loop.run(m, loader, half=False, compute_loss=ComputeLoss(m), nc=nc, profile=True)
    iouv = np.linspace(0.5, 0.95, 10).astype(np.float32)
    stats, losses = [], []
    for im, targets, _, shp in loader:
        with torch.no_grad():
            z, raws = yo.model_forward(cfg, sd, im.float() / 255)[:2]
        losses.append(yo.compute_loss(list(raws), targets.clone(), yo.model_anchors(cfg), hyp=dict(to.HYP), nc=nc)[1].numpy())
        tp_ = targets.numpy().copy()
        tp_[:, 2:] *= 64.0
        out = yo.non_max_suppression(z.numpy(), 0.001, 0.6, multi_label=True, max_det=300)
        for si in range(len(out)):
            lab = tp_[tp_[:, 0] == si, 1:]
            c, _ = yo.val_match_image(out[si], lab, (64, 64), shp[si][0], shp[si][1], iouv)
            stats.append((c, out[si][:, 4], out[si][:, 5], lab[:, 0]))
    tp, conf, pcls, tcls = (np.concatenate(x_, 0) for x_ in zip(*stats))
    assert tp[:, 0].sum() >= 6
    _, _, pp, rr, _, ap, _ = yo.ap_per_class(tp, conf, pcls, tcls)
    # the fp32 HIP forward differs from torch-CPU's in the last bits: a detection at the conf / IoU threshold may flip, so the curve is compared
    # to 2 % rather than to the 1e-12 of the matching kernels' own test
    np.testing.assert_allclose((mp, mr, map50, map_), (pp.mean(), rr.mean(), ap[:, 0].mean(), ap.mean()), rtol=0.02, atol=2e-3)
    np.testing.assert_allclose((lb, lo, lc), np.mean(np.array(losses, dtype=np.float64), 0), rtol=1e-3, atol=1e-5)
    assert all(v >= 0 for v in t)
