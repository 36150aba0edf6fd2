"""CPU: the Python API layer end to end on the host-compiled kernels (tests/hipemu through the `_lib.use_test_library` seam):
model(x) semantics the reference guarantees (fresh result tensors, live weights after a training step, deepcopy / pickle after a
forward), the plan cache, and the guards of the training engine.  References: models/yolo.py:160-170 (forward on live weights),
utils/torch_utils.py:343-365 (ModelEMA deep-copies the model), train.py:469-488 (checkpoint = deepcopy + torch.save)."""
import copy
import io
import os

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo
from tests.hipemu import backend as emu_backend
from yolov5_amd import _state
from yolov5_amd.loss import ComputeLoss
from yolov5_amd.torch_utils import ModelEMA, smart_optimizer
from yolov5_amd.yolo import DetectionModel


@pytest.fixture(autouse=True)
def _seam():
    emu_backend.install()
    yield
    emu_backend.uninstall()


def _model(seed=0):
    m = DetectionModel("yolov5n.yaml")
    m.load_state_dict(yo.det_state_dict(yo.model_cfg("yolov5n"), seed, fused=False))
    m.hyp = dict(yo.HYP_SCRATCH_LOW)
    return m


def _img(seed, B=1, S=64):
    return torch.from_numpy(detgen.uniform((B, 3, S, S), 0.0, 1.0, name="img", seed=seed))


def test_forward_returns_fresh_tensors():
    m = _model().eval()
    a, b = _img(0), _img(1)
    za, raw_a = m(a)
    za_copy = za.clone()
    zb, _ = m(b)
    assert za.data_ptr() != zb.data_ptr()
    assert torch.equal(za, za_copy)          # the second forward did not overwrite the first result (reference semantics)
    assert not torch.equal(za, zb)
    # and against the oracle
    ref = yo.model_forward(yo.model_cfg("yolov5n"), yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False), a)[0]
    np.testing.assert_allclose(za.numpy(), ref.numpy(), rtol=1e-4, atol=2e-4)


def test_eval_after_train_step_uses_live_weights():
    """HipSGD.step_fused, the fused EMA and the train-mode BatchNorm kernel write through raw pointers; the cached
    eval plan must re-pack its filters (running statistics included) instead of replaying stale ones -- for `model` and `ema.ema`."""
    m = _model()
    x = _img(0)
    m.eval()
    z0 = m(x)[0].clone()
    ema = ModelEMA(m)                       # deepcopy after a forward (engine caches must not be copied)
    e0 = ema.ema(x)[0].clone()
    np.testing.assert_allclose(e0.numpy(), z0.numpy(), rtol=1e-5, atol=1e-5)
    opt = smart_optimizer(m, "SGD", lr=0.05, momentum=0.937, decay=5e-4)
    loss_fn = ComputeLoss(m)
    tg = torch.tensor([[0, 3, 0.5, 0.5, 0.3, 0.4], [0, 7, 0.3, 0.6, 0.2, 0.2]], dtype=torch.float32)
    m.train()
    epoch0 = _state.weights_epoch
    for _ in range(2):
        pred = m(x.half())
        loss, _items = loss_fn(pred, tg)
        loss.backward()
        opt.step_fused(inv_scale=1.0, max_norm=10.0, ema=ema, model=m)
        opt.zero_grad()
    assert _state.weights_epoch > epoch0
    m.eval()
    z1 = m(x)[0].clone()                    # same cached plan, refreshed filters
    assert float((z1 - z0).abs().max()) > 1e-4, "eval output did not move after two optimizer steps"
    m.invalidate_engine()
    z1_fresh = m(x)[0]
    assert torch.equal(z1, z1_fresh)        # refreshed plan == plan built from scratch on the new weights
    e1 = ema.ema(x)[0].clone()
    assert float((e1 - e0).abs().max()) > 0
    ema.ema.invalidate_engine()
    assert torch.equal(e1, ema.ema(x)[0])
    # BatchNorm running statistics moved too and are part of the refreshed fold
    bn = next(mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d))
    assert int(bn.num_batches_tracked) == 2


def test_deepcopy_and_pickle_after_forward():
    m = _model().eval()
    x = _img(2)
    m(x)
    m.train()
    m(x.half())                              # training engine cached as well (and the BatchNorm running statistics moved)
    m.eval()
    z = m(x)[0]
    c = copy.deepcopy(m)
    assert "_engine_cache" not in c.__dict__ and "_train_engines" not in c.__dict__
    assert torch.equal(c(x)[0], z)
    buf = io.BytesIO()
    torch.save({"model": copy.deepcopy(m).half(), "ema": None}, buf)   # train.py:469-472
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    zz = ck["model"].float().eval()(x)[0]
    assert float((zz - z).abs().max()) < 0.1   # weights went through fp16


def test_backward_of_a_stale_training_forward_raises():
    m = _model().train()
    x = _img(3).half()
    p1 = m(x)
    p2 = m(x)                                 # overwrites the engine-owned activations p1's backward would read
    with pytest.raises(RuntimeError, match="no longer the model's latest"):
        sum(p.float().sum() for p in p1).backward()
    sum(p.float().sum() for p in p2).backward()   # the latest forward is fine
    assert all(p.grad is not None for p in m.parameters())


def test_plan_cache_keeps_several_shapes(monkeypatch):
    monkeypatch.setenv("Y5_PLAN_CACHE", "2")
    m = _model().eval()
    outs = {}
    for s in (64, 96, 64, 128, 64):
        outs.setdefault(s, []).append(m(_img(4, S=s))[0])
        assert len(m._engines) <= 2
    assert torch.equal(outs[64][0], outs[64][1]) and torch.equal(outs[64][0], outs[64][2])
    assert [k[0][2] for k in m._engines] == [128, 64]   # least recently used shape (96) was dropped


def test_loss_targets_out_of_range():
    m = _model().train()
    loss_fn = ComputeLoss(m)
    p = [torch.zeros(1, 3, s, s, 85) for s in (8, 4, 2)]
    good = torch.tensor([[0, 3, 0.5, 0.5, 0.3, 0.4]])
    bad = torch.tensor([[0, 3, 0.5, 0.5, 0.3, 0.4], [5, 3, 0.5, 0.5, 0.3, 0.4], [0, 99, 0.5, 0.5, 0.3, 0.4]])
    l_good = loss_fn(p, good)[0]
    l_bad = loss_fn(p, bad)[0]                 # rows outside the batch / class list are dropped, nothing is addressed out of range
    assert torch.equal(l_good, l_bad)
    loss_fn.check_targets = True
    with pytest.raises(IndexError):
        loss_fn(p, bad)


def test_autobalance_follows_oracle():
    """ComputeLoss(model, autobalance=True) (utils/loss.py:127, :173-177): the kernels hand each level's objectness loss back through the workspace
    (y5_loss_obji_offset) and the balance list drifts as in the oracle (pinned to the live reference in tests/test_oracle_vs_reference.py); loss AND
    gradient of every call use the factors the call started with."""
    m = _model().train()
    loss_fn = ComputeLoss(m, autobalance=True)
    assert loss_fn.ssi == 1
    anchors = yo.model_anchors(yo.model_cfg("yolov5n"))
    bal = [4.0, 1.0, 0.4]
    for step in range(3):
        pn = [detgen.uniform((2, 3, s, s, 85), -3.0, 3.0, name=f"ab{s}", seed=40 + step) for s in (16, 8, 4)]
        t = torch.from_numpy(detgen.synth_targets(2, 6, seed=40 + step))
        p = [torch.from_numpy(a).clone().requires_grad_(True) for a in pn]
        q = [torch.from_numpy(a).clone().requires_grad_(True) for a in pn]
        loss, items = loss_fn(p, t)
        loss.backward()
        lo, io = yo.compute_loss(q, t, anchors, nc=80, balance=bal, autobalance_ssi=1)
        lo.backward()
        np.testing.assert_allclose(loss.detach().numpy(), lo.detach().numpy(), rtol=2e-6)
        np.testing.assert_allclose(items.numpy(), io.numpy(), rtol=2e-6)
        np.testing.assert_allclose(loss_fn.balance, bal, rtol=1e-9)   # obji differs from torch's mean in the last fp32 bit at most: 1e-4 of that in the EMA
        for a, b in zip(p, q):
            np.testing.assert_allclose(a.grad.numpy(), b.grad.numpy(), rtol=2e-4, atol=2e-9)
    assert loss_fn.balance[1] == 1.0 and loss_fn.balance[0] != 4.0
    with pytest.raises(ValueError):                       # loss.py:127: list(m.stride).index(16) on a head without a stride-16 level
        m.model[-1].stride = torch.tensor([8.0, 32.0, 64.0])
        ComputeLoss(m, autobalance=True)


def test_scale_img_kernel_equals_torch_interpolate_and_pad():
    """utils/torch_utils.py scale_img (F.interpolate bilinear align_corners=False + F.pad 0.447) and the x.flip(3) in front of it (yolo.py:276)."""
    import math

    import torch.nn.functional as F

    from yolov5_amd.torch_utils import scale_img

    x = torch.from_numpy(detgen.uniform((2, 3, 50, 70), 0.0, 1.0, name="si", seed=1))
    for ratio, flip in ((0.83, 3), (0.67, None), (1.0, 3), (0.5, 2), (1.0, None)):
        src = x.flip(flip) if flip else x
        if ratio == 1.0:
            ref = src
        else:
            s = (int(50 * ratio), int(70 * ratio))
            ref = F.interpolate(src, size=s, mode="bilinear", align_corners=False)
            h, w = (math.ceil(v * ratio / 32) * 32 for v in (50, 70))
            ref = F.pad(ref, [0, w - s[1], 0, h - s[0]], value=0.447)
        out = scale_img(x, ratio, gs=32, flip=flip)
        assert out.shape == ref.shape
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-6)
        outh = scale_img(x.half(), ratio, gs=32, flip=flip)
        torch.testing.assert_close(outh.float(), ref, rtol=2e-3, atol=1e-3)


def test_forward_augment_equals_oracle():
    """model(x, augment=True) (models/yolo.py:269-312) on the kernels: three plan runs + y5_scale_img / y5_tta_descale against oracle.forward_augment
    (pinned to the live reference in tests/test_oracle_vs_reference.py)."""
    m = _model().eval()
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 96), 0.0, 1.0, name="tta", seed=5))
    with torch.no_grad():
        z, none = m(x, augment=True)
        zo = yo.forward_augment(yo.model_cfg("yolov5n"), yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False), x)
    assert none is None and z.shape == zo.shape
    np.testing.assert_allclose(z.numpy(), zo.numpy(), rtol=2e-4, atol=5e-4)
    zs = m(x)[0]                                             # the plain forward is untouched by the in-place de-scaling
    np.testing.assert_allclose(zs.numpy(), yo.model_forward(yo.model_cfg("yolov5n"), yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False), x)[0].numpy(),
                               rtol=1e-4, atol=2e-4)


def test_forward_and_nms_under_inference_mode():
    """detect.py / val.py run under `smart_inference_mode` (utils/torch_utils.py:34-43 -> torch.inference_mode).  Inference tensors
    have no version counter; the engine's outputs therefore stay ordinary tensors (the objectness-hint tag keeps working), and NMS of a genuine
    inference tensor (a copy made by the caller) falls back to reading the rows -- same detections either way."""
    from yolov5_amd.general import non_max_suppression

    m = _model().eval()
    x = _img(0, B=2)
    with torch.no_grad():
        z_ref = m(x)[0].clone()
        d_ref = non_max_suppression(z_ref, 0.001, 0.6, max_det=50)
    with torch.inference_mode():
        z = m(x)[0]
        assert not z.is_inference() and getattr(z, "_y5_obj_hint", None) is not None   # ordinary tensor, tagged
        d = non_max_suppression(z, 0.001, 0.6, max_det=50)                              # through the plane
        zc = z.clone()                                                                   # an inference tensor: no version counter
        assert zc.is_inference()
        d2 = non_max_suppression(zc, 0.001, 0.6, max_det=50)
        z[0, 0, 4] = 0.0                                                                 # in-place edit inside inference mode is still seen
        d3 = non_max_suppression(z, 0.001, 0.6, max_det=50)
        d4 = non_max_suppression(z.clone(), 0.001, 0.6, max_det=50)
    assert torch.equal(z_ref, zc)
    assert sum(len(t) for t in d_ref) > 0
    for a, b, c in zip(d_ref, d, d2):
        assert torch.equal(a, b) and torch.equal(a, c)
    for a, b in zip(d3, d4):
        assert torch.equal(a, b)


def test_nms_apriori_labels_equal_oracle():
    """utils/general.py:706-712 `labels` (autolabelling, val.py --save-hybrid) through yolov5_amd.non_max_suppression: label rows appended as extra prediction
    rows; same detections, bit for bit, as the oracle (which is pinned to the live reference for this branch in tests/test_oracle_vs_reference.py),
    for fp32 and fp16 predictions (fp16: the reference's torch.cat promotes to fp32 -- so does this path)."""
    from yolov5_amd.general import non_max_suppression

    pred = detgen.synth_predictions(3, 800, 15, obj_pow=3, seed=43)
    labels = [np.array([[2, 100.0, 120.0, 40.0, 60.0], [7, 300.5, 310.25, 80.0, 20.0]], np.float32), np.zeros((0, 5), np.float32),
              np.array([[0, 50.0, 60.0, 30.0, 30.0]], np.float32)]
    for dt in (torch.float32, torch.float16):
        p = torch.from_numpy(pred).to(dt)
        for ml in (False, True):
            exp = yo.non_max_suppression(p.float().numpy(), 0.25, 0.45, labels=labels, multi_label=ml, max_det=300)
            got = non_max_suppression(p, 0.25, 0.45, labels=[torch.from_numpy(lb) for lb in labels], multi_label=ml, max_det=300)
            for g, e in zip(got, exp):
                assert np.array_equal(g.numpy(), e)
    assert float(got[0][0, 4]) == 1.0


def test_tune_cache_is_bound_to_the_kernel_library(tmp_path, monkeypatch):
    """Tile / split choices are properties of the kernel binary: a cache file written with another build of libyolov5_hip.so is ignored."""
    import json

    from yolov5_amd import engine as eng

    path = tmp_path / "tune.json"
    monkeypatch.setenv("Y5_TUNE_CACHE", str(path))
    saved, state = dict(eng._TUNE_CACHE), dict(eng._TUNE_FILE_STATE)
    try:
        eng._TUNE_CACHE.clear()
        eng._TUNE_FILE_STATE.update(loaded=False)
        eng._TUNE_FILE_STATE.pop("stamp", None)
        eng._TUNE_CACHE[(1, 2, 3)] = (43, 2)
        eng._save_tune_cache()
        d = json.load(open(path))
        assert d["__lib_sha16__"] == eng._lib_stamp() and d["1,2,3"] == [43, 2]
        eng._TUNE_CACHE.clear()
        eng._TUNE_FILE_STATE["loaded"] = False
        eng._load_tune_cache()
        assert eng._TUNE_CACHE == {(1, 2, 3): (43, 2)}
        d["__lib_sha16__"] = "0123456789abcdef"                      # another build
        json.dump(d, open(path, "w"))
        eng._TUNE_CACHE.clear()
        eng._TUNE_FILE_STATE["loaded"] = False
        eng._load_tune_cache()
        assert eng._TUNE_CACHE == {}
    finally:
        eng._TUNE_CACHE.clear(); eng._TUNE_CACHE.update(saved)
        eng._TUNE_FILE_STATE.clear(); eng._TUNE_FILE_STATE.update(state)


def test_shipped_tune_database_seeds_the_choices(tmp_path, monkeypatch):
    """yolov5_amd/tune_db.json (engine._load_tune_cache): taken when it was written for THIS build of the library and no cache file was named explicitly;
    ignored for another build, with Y5_TUNE_CACHE set (tests and profiling scripts keep their own races) and with Y5_TUNE_DB=0; the user's cache wins."""
    import json

    from yolov5_amd import engine as eng

    db = tmp_path / "tune_db.json"
    monkeypatch.setattr(eng, "TUNE_DB_PATH", str(db))
    monkeypatch.setenv("HOME", str(tmp_path))   # the default cache path lives under ~/.cache
    saved, state, stats = dict(eng._TUNE_CACHE), dict(eng._TUNE_FILE_STATE), dict(eng.TUNE_STATS)

    def load(**env):
        for k in ("Y5_TUNE_CACHE", "Y5_TUNE_DB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng._TUNE_CACHE.clear()
        eng._TUNE_FILE_STATE["loaded"] = False
        eng.TUNE_STATS["db_entries"] = 0
        eng._load_tune_cache()
        return dict(eng._TUNE_CACHE)

    try:
        eng._TUNE_FILE_STATE.pop("stamp", None)
        json.dump({"__lib_sha16__": eng._lib_stamp(), "1,2,True": [96, 40], "1,2,True,-777,33": [40, -1]}, open(db, "w"))
        want = {(1, 2, True): (96, 40), (1, 2, True, eng._INSITU_MARK, 33): (40, -1)}
        assert load() == want and eng.TUNE_STATS["db_entries"] == 2
        assert load(Y5_TUNE_DB="0") == {}
        assert load(Y5_TUNE_CACHE=str(tmp_path / "own.json")) == {}
        assert load(Y5_TUNE_CACHE="off") == {}
        # the user's cache (default path; None where the HIP library cannot be loaded, as on a CPU-only host) overrides an entry of the database
        path = eng._tune_cache_path()
        if path:
            import os

            os.makedirs(os.path.dirname(path), exist_ok=True)
            json.dump({"__lib_sha16__": eng._lib_stamp(), "1,2,True": [43, 96]}, open(path, "w"))
            got = load()
            assert got[(1, 2, True)] == (43, 96) and got[(1, 2, True, eng._INSITU_MARK, 33)] == (40, -1)
            os.remove(path)
        json.dump({"__lib_sha16__": "0123456789abcdef", "1,2,True": [96, 40]}, open(db, "w"))   # written for another build of the kernels
        assert load() == {} and eng.TUNE_STATS["db_entries"] == 0
    finally:
        eng._TUNE_CACHE.clear(); eng._TUNE_CACHE.update(saved)
        eng._TUNE_FILE_STATE.clear(); eng._TUNE_FILE_STATE.update(state)
        eng.TUNE_STATS.update(stats)


def test_committed_tune_database_is_well_formed():
    """yolov5_amd/tune_db.json as committed: valid JSON, every key parses into the integer / boolean tuple the cache uses, every value is a pair of integers ((best, runner-up) configuration ids for
    the forward races), the in-situ decisions name one of their race's two candidates.  Whether it is USED is decided at run time by the library hash it carries (a
    database written for other kernels is ignored, test_shipped_tune_database_seeds_the_choices); when the hash matches the library built here that is asserted too."""
    import hashlib
    import json
    import os

    from yolov5_amd import _lib, engine as eng

    if not os.path.isfile(eng.TUNE_DB_PATH):
        pytest.skip("no shipped database")
    d = json.load(open(eng.TUNE_DB_PATH))
    sha = d.pop("__lib_sha16__")
    assert isinstance(sha, str) and len(sha) == 16
    assert len(d) >= 100
    parsed = {}
    for k, v in d.items():
        key = tuple(int(x) if x not in ("True", "False") else x == "True" for x in k.split(","))
        assert isinstance(v, list) and len(v) == 2 and all(isinstance(c, int) for c in v), (k, v)   # (the training plan's entries pack a configuration and a grid size)
        if key[0] >= 0:
            assert all(-1 <= c < 200 for c in v), (k, v)   # forward races: configuration ids
        parsed[key] = tuple(v)
    marks = 0
    for key, v in parsed.items():
        if len(key) > 2 and key[-2] == eng._INSITU_MARK:
            marks += 1
            race = parsed.get(key[:-2])
            assert race is not None and v[0] in race, (key, v, race)
    assert marks >= 20
    if os.path.isfile(_lib.LIB_PATH) and hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16] == sha:
        saved, state = dict(eng._TUNE_CACHE), dict(eng._TUNE_FILE_STATE)
        try:
            eng._TUNE_FILE_STATE.pop("stamp", None)
            assert eng._read_tune_file(eng.TUNE_DB_PATH) == parsed
        finally:
            eng._TUNE_CACHE.clear(); eng._TUNE_CACHE.update(saved)
            eng._TUNE_FILE_STATE.clear(); eng._TUNE_FILE_STATE.update(state)
