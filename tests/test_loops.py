"""CPU (emulator seam): the reference-shaped loops of yolov5_amd/train_loop.py and detect_loop.py.
  * the schedule arithmetic of train.py:234-248,372-434 (accumulate, weight-decay scaling, warm-up interpolation of lr / momentum,
    LambdaLR) is compared exactly with the oracle loop (oracle/train_oracle.py) -- it does not depend on the kernels;
  * a short run of the tiny model: the fp16 HIP loss curve follows the fp32 oracle's (the long run at yolov5n scale, 50 steps, is the
    GPU test tests/test_gpu_loops.py);
  * detect(): letterbox -> forward -> NMS -> scale_boxes against the same steps taken with the oracle's pieces (detect.py:204-248)."""
import copy

import numpy as np
import pytest
import torch

from oracle import detgen, train_oracle as to, yolo_oracle as yo
from oracle.make_golden import TINY_CFG
from tests.hipemu import backend as emu_backend


@pytest.fixture(autouse=True)
def _seam():
    emu_backend.install()
    yield
    emu_backend.uninstall()


def _tiny(seed=0):
    from yolov5_amd.yolo import DetectionModel

    cfg = copy.deepcopy(TINY_CFG)
    spec = yo.state_spec(cfg)
    sd = yo.det_state_dict(cfg, seed, fused=False)
    m = DetectionModel(copy.deepcopy(TINY_CFG))
    m.load_state_dict(sd)
    return m, cfg, sd


def test_lr_lambda_and_scaler_policy():
    from yolov5_amd.train_loop import LossScaler, lr_lambda

    lin, cos = lr_lambda(10, 0.01), lr_lambda(10, 0.01, cos_lr=True)
    assert lin(0) == 1.0 and abs(lin(10) - 0.01) < 1e-12 and abs(lin(5) - 0.505) < 1e-12       # train.py:243-246
    assert cos(0) == 1.0 and abs(cos(10) - 0.01) < 1e-12 and abs(cos(5) - 0.505) < 1e-12       # one_cycle(1, lrf, epochs)
    s = LossScaler(enabled=True, init_scale=1024.0, growth_interval=2)
    s.record(torch.tensor([1.0, 1.0, 1.0, 0.0])); s.update()
    assert s.scale == 512.0 and s.skipped == 1                                                  # overflow: x0.5
    for _ in range(2):
        s.record(torch.tensor([1.0, 1.0, 0.0, 0.0])); s.update()
    assert s.scale == 1024.0                                                                    # 2 clean steps: x2
    assert LossScaler(enabled=False).scale == 1.0


def test_train_loop_schedule_and_loss_curve_vs_oracle():
    from yolov5_amd.train_loop import TensorLoader, train

    m, cfg, sd = _tiny(seed=2)
    imgs, tpi = to.synthetic_set(6, 64, per_img=2, seed=4)
    bs, epochs = 2, 2
    hyp = dict(to.HYP)
    seen = []
    res = train(m, TensorLoader(imgs, tpi, bs), hyp=dict(hyp), epochs=epochs, device="cpu", amp=True,
                on_batch_end=lambda ni, li, opt: seen.append((ni, [g["lr"] for g in opt.param_groups], [g["momentum"] for g in opt.param_groups])))
    ref = to.train_oracle(cfg, sd, imgs, tpi, bs, hyp=dict(hyp), epochs=epochs)
    # schedule: identical numbers (pure host arithmetic)
    np.testing.assert_allclose(np.array(res["lr"]), np.array(ref["lr"]), rtol=1e-12)
    assert len(seen) == 6 and seen[0][0] == 0
    nw = max(round(hyp["warmup_epochs"] * 3), 100)
    assert abs(seen[3][1][0] - np.interp(3, [0, nw], [hyp["warmup_bias_lr"], hyp["lr0"] * ((1 - 1 / epochs) * (1 - hyp["lrf"]) + hyp["lrf"])])) < 1e-12
    assert abs(seen[3][2][0] - np.interp(3, [0, nw], [hyp["warmup_momentum"], hyp["momentum"]])) < 1e-12
    # weight decay scaled by batch_size * accumulate / nbs (train.py:236): 2 * 32 / 64 = 1
    assert abs(res["optimizer"].param_groups[1]["weight_decay"] - hyp["weight_decay"] * bs * round(64 / bs) / 64) < 1e-15
    # accumulate follows np.interp(ni, [0, nw], [1, 32]).round() = 1, 1, 2, 2, 2, 3: the optimizer steps at ni = 0, 1, 3 only
    assert res["ema"].updates == ref["updates"] == 3
    # loss curve: fp16 HIP vs fp32 oracle
    a, b = res["losses"].numpy(), ref["losses"].numpy()
    assert a.shape == b.shape == (6, 3)
    np.testing.assert_allclose(a, b, rtol=0.05, atol=2e-3)
    assert res["scaler"].skipped == 0 and res["scaler"].scale == 65536.0
    # parameters moved the same way (fp16 gradients vs fp32): direction of the total update
    d_h = torch.cat([(p.detach() - sd[k]).flatten() for k, p in m.named_parameters()])
    d_o = torch.cat([(ref["sd"][k] - sd[k]).flatten() for k, _ in m.named_parameters()])
    cosine = float(d_h @ d_o / (d_h.norm() * d_o.norm()))
    assert cosine > 0.9, cosine


def test_host_amp_step_skips_on_overflow_and_feeds_the_scaler():
    """amp=True with a non-fused optimizer (Adam / AdamW / RMSProp of smart_optimizer).  GradScaler.step semantics: an inf / nan
    gradient skips optimizer.step() and halves the scale; finite gradients are unscaled, clipped to max_norm and applied."""
    from yolov5_amd.train_loop import LossScaler, host_amp_step

    w = torch.nn.Parameter(torch.ones(4))
    opt = torch.optim.Adam([w], lr=0.1)
    sc = LossScaler(enabled=True, init_scale=1024.0)
    w.grad = torch.tensor([1024.0, float("inf"), 0.0, 0.0])
    assert host_amp_step(opt, [w], sc, max_norm=10.0) is False
    sc.update()
    assert torch.equal(w.detach(), torch.ones(4)) and sc.scale == 512.0 and sc.skipped == 1
    w.grad = torch.full((4,), 512.0 * 20.0)                      # unscaled: 20 each, norm 40 -> clipped to 10
    assert host_amp_step(opt, [w], sc, max_norm=10.0) is True
    sc.update()
    assert abs(float(w.grad.norm()) - 10.0) < 1e-3 and sc.scale == 512.0 and sc._clean == 1
    assert float((w.detach() - 1.0).abs().max()) > 0.05           # Adam moved the weights


def test_scale_is_frozen_inside_an_accumulation_window(monkeypatch):
    """a pending back-off must not change the scale between two micro-batch backwards of one window (their gradients would be
    unscaled with the wrong factor).  An overflow is injected into the first optimizer step of a run with accumulate = 2 from then on; the
    scale every backward used is recorded."""
    from yolov5_amd import train_loop as tl

    m, cfg, sd = _tiny(seed=2)
    imgs, tpi = to.synthetic_set(8, 64, per_img=2, seed=4)
    used, windows = [], []
    real_record = tl.LossScaler.record

    def record(self, stats):
        if not getattr(self, "_poisoned", False):               # first step reports an overflow
            self._poisoned = True
            stats = torch.tensor([0.0, 0.0, 1.0, 0.0])
        real_record(self, stats)

    monkeypatch.setattr(tl.LossScaler, "record", record)
    real_backward = torch.Tensor.backward

    def backward(self, *a, **k):
        used.append(cur["scaler"].scale)
        return real_backward(self, *a, **k)

    cur = {}
    real_init = tl.LossScaler.__init__

    def init(self, *a, **k):
        real_init(self, *a, **k)
        cur["scaler"] = self

    monkeypatch.setattr(tl.LossScaler, "__init__", init)
    monkeypatch.setattr(torch.Tensor, "backward", backward)
    res = tl.train(m, tl.TensorLoader(imgs, tpi, 2), hyp=dict(to.HYP), epochs=2, device="cpu", amp=True,
                   on_batch_end=lambda ni, li, opt: windows.append(ni))
    assert res["scaler"].skipped == 1
    # ni:      0 | 1 | 2 3 | 4 5 | 6 7   (accumulate = 1, 1, 2, 2, 2, 3 ...: optimizer steps behind ni = 0, 1, 3, 5, ...)
    assert used[0] == 65536.0 and all(u == 32768.0 for u in used[1:]), used
    assert used[2] == used[3] and used[4] == used[5]


def test_distributed_shards_have_equal_batch_counts():
    """129 images over 2 ranks at batch 64 -- without padding rank 0 runs 2 batches and rank 1 one, and HipDDP's all-reduce of the extra
    backward never completes.  SmartDistributedSampler (utils/dataloaders.py:79-103) pads every rank to ceil(n / world) samples."""
    from yolov5_amd.train_loop import TensorLoader, pad_to_common

    imgs = torch.zeros((129, 3, 8, 8), dtype=torch.uint8)
    tpi = [torch.zeros((0, 6)) for _ in range(129)]
    ls = [TensorLoader(imgs, tpi, 64, rank=r, world_size=2) for r in (0, 1)]
    assert len(ls[0]) == len(ls[1]) == 2 and len(ls[0].idx) == len(ls[1].idx) == 65
    assert ls[1].idx[-1] == ls[1].idx[0]                         # padded with the rank's own head, as DistributedSampler does
    assert pad_to_common([5], 7, 4) == [5, 5] and pad_to_common([0, 4], 7, 4) == [0, 4]
    from yolov5_amd.dataloaders import MosaicLoader

    ml = [MosaicLoader([None] * 129, [None] * 129, batch_size=64, rank=r, world_size=2) for r in (0, 1)]
    assert len(ml[0]) == len(ml[1]) == 2


def _val_set(n, hw, seed):
    """n images + labels + the dataloader's `shapes` tuples ((h0, w0), ((gain, gain), (pad_w, pad_h))) for a letterboxed hw x hw batch."""
    imgs, tpi = to.synthetic_set(n, hw, per_img=3, seed=seed)
    geo = [((48, 64), ((1.0, 1.0), (0.0, 8.0))), ((64, 64), ((1.0, 1.0), (0.0, 0.0))), ((128, 96), ((0.5, 0.5), (8.0, 0.0))), ((32, 64), ((1.0, 1.0), (0.0, 16.0)))]
    return imgs, tpi, [geo[i % len(geo)] for i in range(n)]


class _ValLoader:
    def __init__(self, imgs, tpi, shapes, bs):
        self.imgs, self.tpi, self.shapes, self.bs = imgs, tpi, shapes, bs

    def __len__(self):
        return (len(self.imgs) + self.bs - 1) // self.bs

    def __iter__(self):
        for b0 in range(0, len(self.imgs), self.bs):
            ids = list(range(b0, min(b0 + self.bs, len(self.imgs))))
            t = []
            for k, i in enumerate(ids):
                ti = self.tpi[i].clone()
                ti[:, 0] = k
                t.append(ti)
            yield self.imgs[ids], torch.cat(t, 0), [f"img{i}" for i in ids], [self.shapes[i] for i in ids]


def _inside_picks(d, n=3, S=64):
    """Three of the oracle's own detections to serve as labels: boxes whose CENTRE lies inside the image.  (With the objectness saturated the top
    confidences tie to the last bit, so 'rows 0, 5, 11' named different -- sometimes off-image -- boxes on different host CPUs.)"""
    cx, cy = (d[:, 0] + d[:, 2]) / 2, (d[:, 1] + d[:, 3]) / 2
    ok = (cx > 4) & (cx < S - 4) & (cy > 4) & (cy < S - 4)
    idx = np.flatnonzero(ok)
    idx = idx[np.linspace(0, len(idx) - 1, n).astype(int)] if len(idx) >= n else np.arange(n)
    return d[idx]


def test_val_loop_matches_oracle_pipeline():
    """val.py:255-333,390-396 through yolov5_amd.val_loop.run (fp32 model on the emulator) against the same steps taken with the oracle's pieces:
    forward -> NMS(conf 0.001, iou 0.6, multi_label, max_det 300) -> per-image scale_boxes + process_batch -> ap_per_class; plus the validation loss
    (ComputeLoss on the raw head outputs, averaged over batches)."""
    from yolov5_amd import val_loop
    from yolov5_amd.loss import ComputeLoss

    m, cfg, sd = _tiny(seed=3)
    m.hyp = dict(to.HYP)
    imgs, tpi, shapes = _val_set(6, 64, seed=9)
    # head biases up so that the random-init model produces detections over the 0.001 threshold that overlap the labels now and then
    with torch.no_grad():
        for mi in m.model[-1].m:
            mi.bias.view(m.model[-1].na, -1)[:, 4:] += 3.0
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    nc = m.model[-1].nc
    # labels that the model can hit: three of its own (oracle-computed) detections per image, shifted by a pixel or two, so that true positives
    # exist at the low IoU thresholds and fade out towards 0.95
    with torch.no_grad():
        z0 = yo.model_forward(cfg, sd, imgs.float() / 255)[0].numpy()
    d0 = yo.non_max_suppression(z0, 0.001, 0.6, multi_label=True, max_det=300)
    for i, d in enumerate(d0):
        pick = _inside_picks(d)
        xywh = np.stack([(pick[:, 0] + pick[:, 2]) / 2 + 1.0 + i % 2, (pick[:, 1] + pick[:, 3]) / 2 - 1.0, (pick[:, 2] - pick[:, 0]) * 1.05, pick[:, 3] - pick[:, 1]], 1) / 64.0
        tpi[i] = torch.from_numpy(np.concatenate([np.zeros((3, 1), np.float32), pick[:, 5:6], xywh.astype(np.float32)], 1))
    loader = _ValLoader(imgs, tpi, shapes, 4)
    (mp, mr, map50, map_, lb, lo, lc), maps, t = val_loop.run(m, loader, half=False, compute_loss=ComputeLoss(m), nc=nc)
    # oracle pipeline
    iouv = np.linspace(0.5, 0.95, 10)
    stats, losses = [], []
    for im, targets, _, shp in loader:
        x = im.float() / 255 if im.dtype == torch.uint8 else im.float()
        with torch.no_grad():
            z, raws = yo.model_forward(cfg, sd, x)[:2]
        tp_ = targets.numpy().copy()
        losses.append(yo.compute_loss(list(raws), targets.clone(), yo.model_anchors(cfg), hyp=dict(to.HYP), nc=nc)[1].numpy())
        tp_[:, 2:] *= 64.0
        out = yo.non_max_suppression(z.numpy(), 0.001, 0.6, multi_label=True, max_det=300)
        for si in range(len(out)):
            lab = tp_[tp_[:, 0] == si, 1:]
            c, _ = yo.val_match_image(out[si], lab, (64, 64), shp[si][0], shp[si][1], iouv.astype(np.float32))
            stats.append((c, out[si][:, 4], out[si][:, 5], lab[:, 0]))
    tp, conf, pcls, tcls = (np.concatenate(x_, 0) for x_ in zip(*stats))
    assert tp.shape[0] > 50 and tp[:, 0].sum() >= 6 and tp[:, -1].sum() < tp[:, 0].sum()   # a non-trivial precision / recall curve
    _, _, pp, rr, _, ap, cls = yo.ap_per_class(tp, conf, pcls, tcls)
    ref = (pp.mean(), rr.mean(), ap[:, 0].mean(), ap.mean())
    assert ref[2] > 0.05
    np.testing.assert_allclose((mp, mr, map50, map_), ref, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose((lb, lo, lc), np.mean(np.array(losses, dtype=np.float64), 0), rtol=2e-4, atol=1e-6)
    assert maps.shape == (nc,) and len(t) == 3


def test_detect_loop_matches_oracle_pipeline():
    from yolov5_amd.detect_loop import detect

    m, cfg, sd = _tiny(seed=5)
    det = m.model[-1]
    with torch.no_grad():
        for mi in det.m:
            b = mi.bias.view(det.na, -1)
            b[:, 4] += 3.0
            b[:, 5:] += 2.0
    m = m.eval().fuse()
    sd_f = {k: v.detach().clone() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(9)
    ims = [rng.integers(0, 256, (40, 72, 3), dtype=np.uint8), rng.integers(0, 256, (64, 48, 3), dtype=np.uint8),
           rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)]
    out = detect(m, ims, imgsz=64, conf_thres=0.3, iou_thres=0.45, max_det=40, batch_size=2)
    assert len(out) == 3
    for im, got in zip(ims, out):
        lb, ratio, pad = yo.letterbox(im, (64, 64), auto=False)                                        # detect.py's LoadImages
        x = torch.from_numpy(np.ascontiguousarray(lb.transpose(2, 0, 1))[None]).float() / 255            # :205-210
        with torch.no_grad():
            z = yo.model_forward(cfg, sd_f, x)[0]
        e = yo.non_max_suppression(z.numpy(), 0.3, 0.45, max_det=40)[0].copy()                          # :225
        yo.scale_boxes((64, 64), e[:, :4], im.shape[:2])
        e[:, :4] = np.round(e[:, :4])                                                                     # :248
        assert got.shape == e.shape and len(e) > 0
        np.testing.assert_allclose(got.numpy()[:, :4], e[:, :4], atol=1.0)    # a coordinate at x.5 +- 1e-4 may round either way
        assert (np.abs(got.numpy()[:, :4] - e[:, :4]) > 0).mean() < 0.05
        np.testing.assert_allclose(got.numpy()[:, 4:], e[:, 4:], rtol=1e-4, atol=1e-4)


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/data/images"), reason="the reference's sample images are only in the build container")
def test_detect_on_reference_sample_images_C1():
    """BASELINE config C1 (plumbing): data/images/{bus,zidane}.jpg -> detect() end to end (PIL decode, device letterbox, yolov5n
    forward, NMS, scale_boxes + round), small imgsz so the emulator finishes; against the oracle pipeline on the same decoded pixels."""
    import os

    from yolov5_amd.detect_loop import detect, load_image
    from yolov5_amd.yolo import DetectionModel

    cfg = yo.model_cfg("yolov5n")
    sd = yo.det_state_dict(cfg, 0, fused=False)
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(sd)
    det = m.model[-1]
    with torch.no_grad():
        for mi in det.m:
            b = mi.bias.view(det.na, -1)
            b[:, 4] += 1.5
            b[:, 5:] += 1.0
    m = m.eval().fuse()
    sd_f = {k: v.detach().clone() for k, v in m.state_dict().items()}
    paths = [os.path.join("/root/reference/data/images", f) for f in ("bus.jpg", "zidane.jpg")]
    out = detect(m, paths, imgsz=96, conf_thres=0.25, iou_thres=0.45, max_det=100)
    assert len(out) == 2
    from tests import detset

    for pth, got in zip(paths, out):
        im = load_image(pth)
        assert im.ndim == 3 and im.shape[2] == 3 and im.dtype == np.uint8
        lb, _, _ = yo.letterbox(im, (96, 96), auto=False)
        x = torch.from_numpy(np.ascontiguousarray(lb.transpose(2, 0, 1))[None]).float() / 255
        with torch.no_grad():
            z = yo.model_forward(cfg, sd_f, x)[0]
        e = yo.non_max_suppression(z.numpy(), 0.25, 0.45, max_det=100)[0].copy()
        yo.scale_boxes((96, 96), e[:, :4], im.shape[:2])
        e[:, :4] = np.round(e[:, :4])
        assert (got[:, 2] <= im.shape[1]).all() and (got[:, 3] <= im.shape[0]).all() and len(e) > 3
        a = detset.agreement(e, got.numpy(), 0.25, box_atol=1.0, conf_atol=1e-3, margin=1e-3)
        assert a["unmatched_ref"] + a["unmatched_got"] <= 1, a


def test_ema_bn_buffers_keep_tracking_after_a_validation():
    """val.py:187,388 converts the EMA model `.half()` / `.float()` IN PLACE once per epoch; `nn.Module._apply` re-creates
    buffer tensors, so a fused EMA update that cached (model buffer, EMA buffer) tensor pairs kept writing into orphans and the EMA's
    BatchNorm running statistics froze after the first validation.  Two epochs with a validation after each: the EMA's running statistics must
    follow the model's at the LAST optimizer step (ni = 3, in the second epoch), not those of the first epoch."""
    from yolov5_amd.train_loop import TensorLoader, train

    m, cfg, sd = _tiny(seed=2)
    imgs, tpi = to.synthetic_set(6, 64, per_img=2, seed=4)
    vimgs, vtpi, vshapes = _val_set(4, 64, seed=9)
    snaps = {}

    def grab(ni, li, opt):
        snaps[ni] = {k: b.detach().clone() for k, b in m.named_buffers() if k.endswith(("running_mean", "running_var"))}

    res = train(m, TensorLoader(imgs, tpi, 2), hyp=dict(to.HYP), epochs=2, device="cpu", amp=True, on_batch_end=grab,
                val_loader=_ValLoader(vimgs, vtpi, vshapes, 2))
    assert hasattr(res["optimizer"], "step_fused") and res["ema"].updates == 3 and len(res["results"]) == 2
    e_b = dict(res["ema"].ema.named_buffers())
    d = 0.9999 * (1 - np.exp(-3 / 2000))                     # decay of the third update: the EMA is (1 - d) of the model's tensors at ni = 3
    worst_new, worst_old = 0.0, 0.0
    for k, at3 in snaps[3].items():
        scale = float((at3 - snaps[1][k]).abs().max())        # how far the statistics moved between the second and the third optimizer step
        if scale < 1e-6:
            continue
        worst_new = max(worst_new, float((e_b[k].float() - at3).abs().max()) / scale)
        worst_old = max(worst_old, float((e_b[k].float() - snaps[1][k]).abs().max()) / scale)
    assert worst_old > 0.5, worst_old                         # the test can tell the two apart
    assert worst_new < 0.05 + 2 * d, (worst_new, worst_old)   # EMA follows step 3 (one fp16 round trip of the validation allowed for)
