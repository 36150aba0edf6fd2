"""CPU (emulator seam): checkpoints in the reference's format, both directions, and the inference wrappers built on them.
  * tests/golden/ckpt_ref_tiny.pt was written by the UNMODIFIED reference (oracle/make_golden.py:gen_ckpt, train.py:469-488 layout,
    pickled `models.yolo.*` / `models.common.*` classes): `attempt_load` (models/experimental.py:60-101) must unpickle it into the
    yolov5_amd classes and reproduce the reference's own outputs (ckpt_ref_tiny.npz);
  * `save_checkpoint` writes yolov5_amd models under the reference's class paths: reloaded here, and -- where /root/reference
    exists -- unpickled by the real reference in a subprocess whose forward must agree with ours;
  * DetectMultiBackend (common.py:456-490,685-694), AutoShape + Detections (common.py:843-1101) run on top of the loaded model."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import detgen, ref_shim, yolo_oracle as yo
from tests.hipemu import backend as emu_backend

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CKPT = os.path.join(G, "ckpt_ref_tiny.pt")


@pytest.fixture(autouse=True)
def _seam():
    emu_backend.install()
    yield
    emu_backend.uninstall()


def _x():
    return torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=11))


def test_attempt_load_reference_checkpoint():
    from yolov5_amd import common, yolo
    from yolov5_amd.experimental import attempt_load

    g = np.load(os.path.join(G, "ckpt_ref_tiny.npz"))
    m = attempt_load(CKPT, device="cpu")
    assert type(m) is yolo.DetectionModel and type(m.model[0]) is common.Conv and type(m.model[-1]) is yolo.Detect
    assert not m.training and not hasattr(m.model[0], "bn")          # fused + eval (experimental.py:88)
    assert m.names[3] == "class3" and float(m.stride.max()) == 32.0
    assert sum(p.numel() for p in m.parameters()) == int(g["nparams"])
    z, raw = m(_x())
    np.testing.assert_allclose(z.numpy(), g["z"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(raw[0].numpy(), g["raw0"], rtol=1e-4, atol=2e-4)


def test_detect_multibackend_pt_branch():
    from yolov5_amd.common import DetectMultiBackend

    g = np.load(os.path.join(G, "ckpt_ref_tiny.npz"))
    dmb = DetectMultiBackend(CKPT, device=torch.device("cpu"), fp16=False, fuse=True)
    assert (dmb.pt, dmb.jit, dmb.onnx, dmb.engine, dmb.xml, dmb.triton) == (True, False, False, False, False, False)
    assert dmb.stride == 32 and dmb.names[0] == "class0" and dmb.fp16 is False and dmb.device.type == "cpu"
    dmb.warmup(imgsz=(1, 3, 64, 64))
    y = dmb(_x())
    np.testing.assert_allclose(y[0].numpy(), g["z"], rtol=1e-4, atol=2e-4)
    half = DetectMultiBackend(CKPT, device=torch.device("cpu"), fp16=True)
    zh = half(_x())[0]                       # fp32 input is cast like common.py:688-689
    assert zh.dtype == torch.float16 and float((zh.float() - torch.from_numpy(g["z"])).abs().max()) < 0.5


def test_autoshape_detections_against_oracle_pipeline():
    """AutoShape (letterbox -> model -> NMS -> scale_boxes, common.py:896-946) on two numpy images of different sizes vs the same
    pipeline spelled out with the oracle's pieces; then the Detections accessors (common.py:950-1000)."""
    from yolov5_amd.common import AutoShape, Detections, DetectMultiBackend

    dmb = DetectMultiBackend(CKPT, device=torch.device("cpu"))
    det = dmb.model.model[-1]
    with torch.no_grad():   # a head that fires: objectness / class biases up
        for mi in det.m:
            b = mi.bias.view(det.na, -1)
            b[:, 4] += 3.0
            b[:, 5:] += 2.0
    dmb.model.invalidate_engine()
    auto = AutoShape(dmb, verbose=False)
    auto.conf, auto.max_det = 0.3, 50
    rng = np.random.default_rng(5)
    ims = [rng.integers(0, 256, (48, 80, 3), dtype=np.uint8), rng.integers(0, 256, (96, 64, 3), dtype=np.uint8)]
    res = auto(ims, size=64)
    assert isinstance(res, Detections) and len(res) == 2 and res.s[0] == 2
    # --- the same with oracle parts
    shape1 = tuple(res.s[2:])
    assert shape1 == (64, 64)
    batch = [yo.letterbox(im, shape1, auto=False)[0] for im in ims]                                    # common.py:922
    xb = torch.from_numpy(np.ascontiguousarray(np.stack(batch).transpose(0, 3, 1, 2))).float() / 255     # :923-926
    with torch.no_grad():
        zz = dmb.model(xb)[0]
    exp = yo.non_max_suppression(zz.numpy(), 0.3, 0.45, max_det=50)                                     # :936-939
    for i, (e, p) in enumerate(zip(exp, res.pred)):
        e = e.copy()
        yo.scale_boxes(shape1, e[:, :4], ims[i].shape[:2])                                              # :940-941
        assert len(e) == len(p) and len(e) > 0
        np.testing.assert_allclose(p.numpy(), e, rtol=1e-5, atol=1e-3)
    for i, im in enumerate(ims):
        p = res.pred[i]
        assert p.shape[1] == 6 and (p[:, 0] >= 0).all() and (p[:, 2] <= im.shape[1]).all() and (p[:, 3] <= im.shape[0]).all()
        gn = torch.tensor([im.shape[1], im.shape[0], im.shape[1], im.shape[0], 1, 1], dtype=torch.float32)
        assert torch.equal(res.xyxy[i], p) and torch.allclose(res.xyxyn[i], p / gn)
        xywh = res.xywh[i]
        assert torch.allclose(xywh[:, 0], (p[:, 0] + p[:, 2]) / 2) and torch.allclose(xywh[:, 2], p[:, 2] - p[:, 0])
        assert torch.allclose(res.xywhn[i], xywh / gn)
    assert "image 1/2: 48x80" in str(res) and len(res.tolist()) == 2


def _small_model():
    from oracle.make_golden import TINY_CFG
    from yolov5_amd.yolo import DetectionModel

    torch.manual_seed(3)
    m = DetectionModel(dict(TINY_CFG))
    m.hyp = dict(yo.HYP_SCRATCH_LOW)
    m.names = {i: str(i) for i in range(80)}
    return m


def test_save_checkpoint_round_trip_and_resume(tmp_path):
    from yolov5_amd import yolo
    from yolov5_amd.checkpoint import load_checkpoint, save_checkpoint, smart_resume
    from yolov5_amd.experimental import attempt_load
    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.torch_utils import ModelEMA, smart_optimizer

    m = _small_model()
    x = _x()
    opt = smart_optimizer(m, "SGD", lr=0.02, momentum=0.937, decay=5e-4)
    ema = ModelEMA(m)
    loss_fn = ComputeLoss(m)
    tg = torch.tensor([[0, 3, 0.5, 0.5, 0.3, 0.4], [1, 7, 0.3, 0.6, 0.2, 0.2]])
    m.train()
    for _ in range(2):
        loss, _ = loss_fn(m(x.half()), tg)
        loss.backward()
        opt.step_fused(max_norm=10.0, ema=ema, model=m)
        opt.zero_grad()
    path = tmp_path / "last.pt"
    save_checkpoint(path, m, ema=ema, optimizer=opt, epoch=4, best_fitness=0.5, opt={"imgsz": 64})
    assert yolo.DetectionModel.__module__ == "yolov5_amd.yolo"        # class paths restored after the save
    raw = open(path, "rb").read()
    assert b"models.yolo" in raw and b"yolov5_amd" not in raw          # the pickle names reference paths only
    ck = load_checkpoint(path)
    assert set(ck) >= {"epoch", "best_fitness", "model", "ema", "updates", "optimizer", "opt", "date"} and ck["epoch"] == 4
    assert next(ck["model"].parameters()).dtype == torch.float16 and type(ck["ema"]) is yolo.DetectionModel
    # inference from the file == inference from the live EMA (through fp16 weights)
    ref = ema.ema(x)[0]
    got = attempt_load(path, device="cpu")(x)[0]                       # attempt_load prefers ckpt["ema"] (experimental.py:74)
    assert float((got - ref).abs().max()) < 0.25 and float((got[..., 4:] - ref[..., 4:]).abs().max()) < 5e-3
    # resume: fresh model / optimizer / EMA pick up where the file left off (train.py:218-221, torch_utils.py:293-312)
    m2 = _small_model()
    m2.load_state_dict(ck["model"].float().state_dict())
    opt2 = smart_optimizer(m2, "SGD", lr=0.02, momentum=0.937, decay=5e-4)
    ema2 = ModelEMA(m2)
    best, start, epochs = smart_resume(ck, opt2, ema2, weights=str(path), epochs=10, resume=True)
    assert (best, start, epochs) == (0.5, 5, 10) and ema2.updates == ema.updates == 2
    mb = [opt.state[p]["momentum_buffer"] for g in opt.param_groups for p in g["params"]]
    mb2 = [opt2.state[p]["momentum_buffer"] for g in opt2.param_groups for p in g["params"]]
    assert len(mb) == len(mb2) and all(torch.equal(a, b) for a, b in zip(mb, mb2))
    for (k, a), (_, b) in zip(ema.ema.state_dict().items(), ema2.ema.state_dict().items()):
        if a.dtype.is_floating_point:
            assert float((a - b).abs().max()) <= 1e-3 * (1 + float(a.abs().max())), k   # through fp16
    # strip_optimizer (general.py:770-787): EMA becomes the model, bookkeeping cleared, fp16, no grad; still reference class paths
    from yolov5_amd.checkpoint import strip_optimizer

    best = tmp_path / "best.pt"
    mb_size = strip_optimizer(path, best)
    sk = load_checkpoint(best)
    assert mb_size > 0 and sk["epoch"] == -1 and all(sk[k] is None for k in ("optimizer", "best_fitness", "ema", "updates"))
    assert all(p.dtype == torch.float16 and not p.requires_grad for p in sk["model"].parameters())
    assert b"yolov5_amd" not in open(best, "rb").read()
    for (k, a), (_, b) in zip(ck["ema"].state_dict().items(), sk["model"].state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.skipif(not ref_shim.available(), reason="needs the reference tree (build container only)")
def test_our_checkpoint_loads_in_the_unmodified_reference(tmp_path):
    """The file save_checkpoint writes is a valid ultralytics/yolov5 checkpoint: a separate process imports the REAL reference
    (oracle/ref_shim), torch.load()s the file -- which now resolves to the reference's own classes -- and runs its own forward."""
    from yolov5_amd.checkpoint import save_checkpoint

    m = _small_model().eval()
    x = _x()
    path = tmp_path / "ours.pt"
    save_checkpoint(path, m, epoch=0)
    ours = m.half().float()(x)[0].numpy()    # the file holds fp16 weights
    np.save(tmp_path / "ours.npy", ours)
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from oracle import ref_shim, detgen
ns = ref_shim.load()
ck = torch.load({str(path)!r}, map_location="cpu", weights_only=False)
mm = ck["model"]
assert type(mm).__module__ == "models.yolo" and type(mm) is ns.yolo.DetectionModel, type(mm)
assert type(mm.model[0]) is ns.common.Conv
mm = mm.float().fuse().eval()
x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=11))
with torch.no_grad():
    z = mm(x)[0].numpy()
ours = np.load({str(tmp_path / 'ours.npy')!r})
err = float(np.abs(z - ours).max())
assert err < 2e-3, err
print("REF_LOADED_OK", err)
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "REF_LOADED_OK" in out.stdout, out.stdout + out.stderr
