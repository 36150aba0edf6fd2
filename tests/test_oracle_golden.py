"""CPU: the oracle restatement (oracle/yolo_oracle.py) reproduces the committed golden vectors that
oracle/make_golden.py generated from the unmodified reference.  Travels to the GPU box (no /root/reference)."""
import os

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo
from oracle.make_golden import NMS_CASES, loss_case, nms_case_pred

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("name,model,hw,bs,seed", [("yolov5n_64", "yolov5n", 64, 2, 0), ("yolov5s_320", "yolov5s", 320, 2, 1),
                                                   ("yolov5n-seg_64", "yolov5n-seg", 64, 2, 2)])
def test_forward_matches_reference_golden(name, model, hw, bs, seed):
    g = _load(f"fwd_{name}.npz")
    cfg = yo.model_cfg(model)
    x = torch.from_numpy(detgen.uniform((bs, 3, hw, hw), 0.0, 1.0, name="img", seed=seed))
    assert np.array_equal(g["anchors"], yo.model_anchors(cfg).numpy())
    assert np.array_equal(g["stride"], yo.model_strides(cfg).numpy())
    for fused in (False, True):
        sd = yo.det_state_dict(cfg, seed, fused=fused)
        with torch.no_grad():
            out = yo.model_forward(cfg, sd, x)
        z = out[0].numpy()
        key = "z_fused" if fused else "z_unfused"
        rs = int(g["row_stride"])
        if rs == 1:
            np.testing.assert_allclose(z, g[key], rtol=2e-4, atol=2e-4)
            if not fused:
                raws = out[2] if "seg" in model else out[1]
                for i, r in enumerate(raws):
                    np.testing.assert_allclose(r.numpy(), g[f"raw{i}"], rtol=2e-4, atol=2e-4)
        else:
            np.testing.assert_allclose(z.reshape(-1, z.shape[-1])[::rs], g[key + "_rows"], rtol=2e-4, atol=5e-4)
            s = z.astype(np.float64)
            np.testing.assert_allclose([s.sum(), np.abs(s).sum(), (s * s).sum()], g[key + "_sum"], rtol=1e-5)
        if "seg" in model and fused:
            proto = out[1].numpy()
            np.testing.assert_allclose(proto[:, :, ::5, ::5], g["proto_sample"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("name", ["yolov5s_640", "yolov5x_1280", "yolov5s-seg_640"])
def test_full_resolution_forward_and_nms_match_reference_golden(name):
    """BASELINE configs C2 / C4 / C5 at their real resolution: the oracle's fused forward and NMS against the reference's own
    (tests/golden/detset_*.npz): z within 2e-4, detections identical in count, order and class, boxes / confidences within 1e-3."""
    from tests import detset

    g, cfg, x, seed, seg = detset.load(name)
    sd = detset.state_dict(name, g, fused=True)
    with torch.no_grad():
        out = yo.model_forward(cfg, sd, x)
    z = out[0].numpy()
    assert tuple(g["shape"]) == z.shape
    rs = int(g["row_stride"])
    np.testing.assert_allclose(z.reshape(-1, z.shape[-1])[::rs], g["z_rows"], rtol=2e-4, atol=1e-3)
    s = z.astype(np.float64)
    np.testing.assert_allclose([s.sum(), np.abs(s).sum(), (s * s).sum()], g["z_sum"], rtol=1e-5)
    if seg:
        np.testing.assert_allclose(out[1].numpy()[:, :, ::5, ::5], g["proto_sample"], rtol=2e-4, atol=2e-4)
    conf, iou, max_det = float(g["nms"][0]), float(g["nms"][1]), int(g["nms"][2])
    dets = yo.non_max_suppression(z, conf, iou, max_det=max_det, nm=32 if seg else 0)
    for i, d in enumerate(dets):
        ref = g[f"det{i}"]
        a = detset.agreement(ref, d, conf, box_atol=0.05, conf_atol=1e-3, margin=1e-3)
        assert a["unmatched_ref"] == 0 and a["unmatched_got"] == 0 and abs(len(ref) - len(d)) <= 2, (name, i, a, len(ref), len(d))


def test_letterbox_restatement_matches_reference_golden():
    from oracle.make_golden import LETTERBOX_CASES, LETTERBOX_GEOMETRY, letterbox_image

    g = _load("letterbox.npz")
    for name, (_, kw) in LETTERBOX_CASES.items():
        im, ratio, pad = yo.letterbox(letterbox_image(name), **kw)
        assert np.array_equal(im, g[name]), name
        np.testing.assert_allclose([ratio[0], ratio[1], pad[0], pad[1]], g[name + "_meta"], rtol=0, atol=1e-12)
    for row, ((h, w), kw) in zip(g["geometry"], LETTERBOX_GEOMETRY[::7]):
        pass  # (geometry rows are checked against the product's host geometry in tests/test_emu_letterbox.py)
    for k, ((h, w), kw) in enumerate(LETTERBOX_GEOMETRY):
        if k % 7:
            continue
        im, ratio, pad = yo.letterbox(np.zeros((h, w, 3), np.uint8), **kw)
        np.testing.assert_allclose([im.shape[0], im.shape[1], ratio[0], ratio[1], pad[0], pad[1]], g["geometry"][k][:6], atol=1e-12)


def test_nparams_match():
    g = _load("fwd_yolov5s_320.npz")
    spec = yo.state_spec(yo.model_cfg("yolov5s"))
    n = sum(int(np.prod(v.shape)) for k, v in spec.items()
            if not k.endswith(("running_mean", "running_var", "num_batches_tracked", "anchors")))
    assert n == int(g["nparams"]) == 7235389  # yolov5s params


def test_detect_grid_bit_exact():
    g = _load("fwd_yolov5n_64.npz")
    cfg = yo.model_cfg("yolov5n")
    anchors, strides = yo.model_anchors(cfg), yo.model_strides(cfg)
    for i, s in enumerate((8, 4, 2)):
        grid, ag = yo.make_grid(s, s, anchors[i], strides[i])
        assert np.array_equal(grid[0, 0].numpy(), g[f"grid{i}"])
        assert np.array_equal(ag[0, :, 0, 0].numpy(), g[f"anchor_grid{i}"])


def test_fuse_conv_bn():
    g = _load("fuse.npz")
    w, b = yo.fuse_conv_and_bn(torch.from_numpy(detgen.uniform((16, 8, 3, 3), -0.5, 0.5, name="fw")),
                               torch.from_numpy(detgen.uniform((16,), 0.5, 1.5, name="fg")),
                               torch.from_numpy(detgen.uniform((16,), -0.5, 0.5, name="fb")),
                               torch.from_numpy(detgen.uniform((16,), -0.5, 0.5, name="fm")),
                               torch.from_numpy(detgen.uniform((16,), 0.5, 1.5, name="fv")))
    np.testing.assert_allclose(w.numpy(), g["w"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b.numpy(), g["b"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", list(NMS_CASES))
def test_nms_bit_exact(name):
    g = _load("nms.npz")
    kw, nkw = NMS_CASES[name]
    out = yo.non_max_suppression(nms_case_pred(name), **nkw)
    for i, o in enumerate(out):
        ref = g[f"{name}_{i}"]
        assert o.shape == ref.shape, (name, i, o.shape, ref.shape)
        assert np.array_equal(o, ref), (name, i)


@pytest.mark.parametrize("name", ["appendix_a", "synthetic", "no_targets"])
def test_loss_and_targets(name):
    g = _load("loss.npz")
    pn, tn = loss_case(name)
    anchors = torch.from_numpy(g["anchors"])
    assert np.array_equal(g["anchors"], yo.model_anchors(yo.model_cfg("yolov5s")).numpy())
    p = [torch.from_numpy(a).clone().requires_grad_(True) for a in pn]
    t = torch.from_numpy(tn)
    tcls, tbox, indices, anch = yo.build_targets([q.shape for q in p], t, anchors)
    for i in range(3):
        idx = torch.stack(indices[i]).numpy() if indices[i][0].numel() else np.zeros((4, 0), np.int64)
        assert np.array_equal(idx, g[f"{name}_idx{i}"])  # int64 indices: bit-exact
        assert np.array_equal(tcls[i].numpy(), g[f"{name}_tcls{i}"])
        assert np.array_equal(tbox[i].numpy(), g[f"{name}_tbox{i}"])
        assert np.array_equal(anch[i].numpy(), g[f"{name}_anch{i}"])
    loss, items = yo.compute_loss(p, t, anchors)
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), g[f"{name}_loss"], rtol=1e-6)
    np.testing.assert_allclose(items.numpy(), g[f"{name}_items"], rtol=1e-6, atol=1e-8)
    for i in range(3):
        gr = p[i].grad.numpy()
        if f"{name}_grad{i}" in g:
            np.testing.assert_allclose(gr, g[f"{name}_grad{i}"], rtol=1e-5, atol=1e-9)
        else:
            s = gr.astype(np.float64)
            np.testing.assert_allclose([s.sum(), np.abs(s).sum()], g[f"{name}_grad{i}_sum"], rtol=1e-6)
            nz = g[f"{name}_grad{i}_nzidx"]
            if len(nz):
                np.testing.assert_allclose(gr[tuple(nz.T)], g[f"{name}_grad{i}_nzrows"], rtol=1e-5, atol=1e-9)


def test_appendix_a_known_answer():
    """lobj = ln2*(4+1+0.4)*hyp.obj, lcls = ln2*3*hyp.cls at zero logits (no third-party code)."""
    pn, tn = loss_case("appendix_a")
    anchors = yo.model_anchors(yo.model_cfg("yolov5s"))
    _, items = yo.compute_loss([torch.from_numpy(a) for a in pn], torch.from_numpy(tn), anchors)
    assert abs(items[1].item() - np.log(2) * 5.4) < 1e-5
    assert abs(items[2].item() - np.log(2) * 1.5) < 1e-5
    assert abs(items[0].item() - 0.09738069) < 1e-6


def test_process_mask():
    g = _load("mask.npz")
    protos = torch.from_numpy(detgen.uniform((32, 40, 40), -1.0, 1.0, name="protos", seed=13))
    coef = torch.from_numpy(detgen.uniform((7, 32), -1.0, 1.0, name="coef", seed=13))
    xy1 = detgen.uniform((7, 2), 0, 90, name="bx1", seed=13)
    wh = detgen.uniform((7, 2), 8, 70, name="bwh", seed=13)
    boxes = torch.from_numpy(np.concatenate((xy1, xy1 + wh), 1))
    for up, key in ((False, "noup"), (True, "up")):
        m = yo.process_mask(protos, coef, boxes, (160, 160), upsample=up).numpy().astype(bool)
        ref = np.unpackbits(g[f"m_{key}"])[: m.size].reshape(g[f"shape_{key}"]).astype(bool)
        assert (m != ref).mean() < 1e-4


def test_scale_boxes():
    g = _load("scale_boxes.npz")
    b = detgen.uniform((20, 4), -20, 660, name="sb", seed=14)
    np.testing.assert_allclose(yo.scale_boxes((640, 640), b.copy(), (1080, 810)), g["a"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(yo.scale_boxes((384, 640), b.copy(), (720, 1280)), g["b"], rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_training_sample_pipeline_matches_reference_golden(seed):
    """oracle/augment_oracle.py (mosaic + random_perspective + HSV + flips + CHW/RGB + collate, draws passed in) against the batch the
    reference's own LoadImagesAndLabels.__getitem__ / collate_fn produced (tests/golden/augment.npz): pixels and labels identical."""
    from oracle import augment_oracle as ao

    g = _load("augment.npz")
    s = int(g["s"])
    ims, labs = ao.synthetic_dataset(6, seed=3)
    labs = [lb.astype(np.float32) for lb in labs]
    hyp = dict(ao.HYP_AUG, degrees=5.0, shear=2.0, flipud=0.3)
    samples = []
    for index in (seed % 6, (seed + 3) % 6):
        d = ao.reference_draws(seed * 10 + index, index, 6, s, hyp)
        samples.append(ao.mosaic_sample(ims, labs, d, s, hyp))
    imb, labb = ao.collate(samples)
    assert np.array_equal(imb, g[f"img{seed}"])
    assert labb.shape == g[f"lab{seed}"].shape and np.array_equal(labb, g[f"lab{seed}"])


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_training_sample_pipeline_mixed_branches_match_reference_golden(seed):
    """hyp['mosaic'] = 0.5: the gate of dataloaders.py:701 sends samples down the letterbox branch (:710-733: load_image, letterbox, float-pad
    labels, random_perspective with border (0, 0)) or the mosaic branch; oracle restatement of both against the reference's own
    __getitem__ / collate_fn (tests/golden/augment_mixed.npz), branch by branch as the reference's generator decided."""
    from oracle import augment_oracle as ao

    g = _load("augment_mixed.npz")
    s = int(g["s"])
    ims, labs = ao.synthetic_dataset(6, seed=3)
    labs = [lb.astype(np.float32) for lb in labs]
    hyp = dict(ao.HYP_AUG, degrees=5.0, shear=2.0, flipud=0.3, mosaic=0.5)
    samples, gates = [], []
    for index in (seed % 6, (seed + 2) % 6, (seed + 4) % 6):
        d = ao.reference_draws(seed * 10 + index, index, 6, s, hyp)
        gates.append(d["mosaic"])
        samples.append(ao.sample(ims, labs, d, s, hyp))
    assert gates == list(g[f"mosaic{seed}"])
    imb, labb = ao.collate(samples)
    assert np.array_equal(imb, g[f"img{seed}"])
    assert labb.shape == g[f"lab{seed}"].shape and np.array_equal(labb, g[f"lab{seed}"])


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_training_sample_pipeline_mixup_matches_reference_golden(seed):
    """hyp['mixup'] = 0.5 (dataloaders.py:707-708 -> utils/augmentations.py:225-233): the oracle's restatement -- partner mosaic with its own draws,
    float64 blend truncated to uint8, labels concatenated -- against the reference's own __getitem__ / collate_fn (tests/golden/augment_mixup.npz)."""
    from oracle import augment_oracle as ao

    g = _load("augment_mixup.npz")
    s = int(g["s"])
    ims, labs = ao.synthetic_dataset(6, seed=3)
    labs = [lb.astype(np.float32) for lb in labs]
    hyp = dict(ao.HYP_AUG, degrees=5.0, shear=2.0, flipud=0.3, mixup=0.5)
    samples = [ao.sample(ims, labs, ao.reference_draws(seed * 10 + index, index, 6, s, hyp), s, hyp) for index in (seed % 6, (seed + 2) % 6, (seed + 4) % 6)]
    imb, labb = ao.collate(samples)
    assert np.array_equal(imb, g[f"img{seed}"])
    assert labb.shape == g[f"lab{seed}"].shape and np.array_equal(labb, g[f"lab{seed}"])


def test_loss_focal():
    """hyp fl_gamma = 1.5 + label smoothing 0.1 (utils/loss.py:120-122 -> FocalLoss :77-98): the oracle's restatement against the reference's own
    ComputeLoss (tests/golden/loss_focal.npz, oracle/make_golden.py gen_loss)."""
    g, gp = _load("loss_focal.npz"), _load("loss.npz")
    pn, tn = loss_case("synthetic")
    p = [torch.from_numpy(a).clone().requires_grad_(True) for a in pn]
    hyp = dict(yo.HYP_SCRATCH_LOW, fl_gamma=1.5, label_smoothing=0.1)
    loss, items = yo.compute_loss(p, torch.from_numpy(tn), torch.from_numpy(gp["anchors"]), hyp=hyp)
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose(items.numpy(), g["items"], rtol=1e-6)
    for i in range(3):
        gr = p[i].grad.numpy()
        s = gr.astype(np.float64)
        np.testing.assert_allclose([s.sum(), np.abs(s).sum()], g[f"grad{i}_sum"], rtol=1e-6)
        nz = g[f"grad{i}_nzidx"]
        np.testing.assert_allclose(gr[tuple(nz.T)], g[f"grad{i}_nzrows"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(gr[0, 0, :4, :8, 4], g[f"grad{i}_obj_head"], rtol=1e-5, atol=1e-10)
