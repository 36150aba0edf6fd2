"""CPU (no GPU): whole-model check of the planner + plan materialisation + kernels on the HIP emulator:
yolov5n / yolov5n-seg at 64x64 against the golden vectors generated from the unmodified reference."""
import os

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo
from tests.hipemu.backend import EmuBackend
from yolov5_amd.engine import Engine, build_plan_spec
from yolov5_amd.yolo import DetectionModel, SegmentationModel

G = os.path.join(os.path.dirname(__file__), "golden")


def det_model(name, seed, fused):
    M = SegmentationModel if "seg" in name else DetectionModel
    m = M(name + ".yaml")
    sd = yo.det_state_dict(yo.model_cfg(name), seed, fused=False)
    m.load_state_dict(sd)
    m.eval()
    if fused:
        m.fuse()
    return m


@pytest.mark.parametrize("fused", [False, True])
def test_yolov5n_fp32_matches_reference_golden(fused):
    g = np.load(os.path.join(G, "fwd_yolov5n_64.npz"))
    m = det_model("yolov5n", 0, fused)
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=0))
    eng = Engine(m, (2, 3, 64, 64), torch.float32, "cpu", want_raw=True, backend=EmuBackend())
    out = eng(x)
    ref = g["z_fused" if fused else "z_unfused"]
    # north_star tolerance: fp32 boxes within 1e-4 (relative to the 64 px image here; values up to ~64)
    np.testing.assert_allclose(out["z"], ref, rtol=1e-4, atol=1e-4)
    for i in range(3):
        np.testing.assert_allclose(out[f"raw{i}"], g[f"raw{i}"], rtol=1e-4, atol=2e-4)


def test_yolov5n_fp16_close_to_reference():
    g = np.load(os.path.join(G, "fwd_yolov5n_64.npz"))
    m = det_model("yolov5n", 0, True).half()
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=0)).half()
    eng = Engine(m, (2, 3, 64, 64), torch.float16, "cpu", want_raw=False, backend=EmuBackend())
    z = eng(x)["z"].astype(np.float32)
    assert eng._stem is not None and eng._stem_active  # fp16 NCHW batch -> fused stem kernel (conv_stem.h), no repack pass
    ref = g["z_fused"]
    # check_amp-style tolerance (utils/general.py:410-435 uses atol=0.1 on boxes) -- fp16 storage between layers
    assert np.abs(z - ref).max() < 0.5
    assert np.abs(z[..., 4:] - ref[..., 4:]).max() < 2e-2


def test_yolov5n_seg_fp32():
    g = np.load(os.path.join(G, "fwd_yolov5n-seg_64.npz"))
    m = det_model("yolov5n-seg", 2, True)
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=2))
    eng = Engine(m, (2, 3, 64, 64), torch.float32, "cpu", want_raw=False, backend=EmuBackend())
    out = eng(x)
    np.testing.assert_allclose(out["z"], g["z_fused"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(out["proto"][:, :, ::5, ::5], g["proto_sample"], rtol=1e-4, atol=1e-4)


def test_plan_is_concat_free_for_shipped_models():
    for name in ("yolov5s", "yolov5x", "yolov5s-seg"):
        M = SegmentationModel if "seg" in name else DetectionModel
        spec = build_plan_spec(M(name + ".yaml").eval(), 1, 3, 640, 640)
        kinds = [o["op"] for o in spec.ops]
        assert "copy" not in kinds and "upsample" not in kinds  # every concat / upsample is a fused store
        assert kinds.count("sppf_pool") == 1 and kinds.count("decode") == 3


def test_head_side_branch_schedule(monkeypatch):
    """engine._Planner._schedule_heads: the Detect heads of the lower pyramid levels (and the Segment proto branch) sit right behind
    the op that completes their input and are marked for the plan's side stream; same ops, same results as the single-stream order."""
    for name, M in (("yolov5n", DetectionModel), ("yolov5n-seg", SegmentationModel)):
        m = det_model(name, 0, True)
        monkeypatch.setenv("Y5_EXPERIMENTAL", "")
        flat = build_plan_spec(m, 2, 3, 64, 64)
        monkeypatch.setenv("Y5_EXPERIMENTAL", "head_branch")
        br = build_plan_spec(m, 2, 3, 64, 64)
        key = lambda o: (o["op"], o.get("name"), o.get("level"))  # noqa: E731
        assert sorted(map(key, flat.ops), key=str) == sorted(map(key, br.ops), key=str)
        assert not any(o.get("side") for o in flat.ops)
        side = [i for i, o in enumerate(br.ops) if o.get("side")]
        assert side and all(br.ops[i]["head"] != 2 for i in side)  # the last level stays on the main stream
        for i in side:
            if not br.ops[i - 1].get("side"):  # fork point: the previous op wrote this op's input
                prev, x = br.ops[i - 1], br.ops[i]["x"]
                assert prev["y"].buf == x.buf
        x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=0))
        a = Engine(m, (2, 3, 64, 64), torch.float32, "cpu", backend=EmuBackend(), spec=flat)(x)
        b = Engine(m, (2, 3, 64, 64), torch.float32, "cpu", backend=EmuBackend(), spec=br)(x)
        for k in a:
            assert np.array_equal(a[k], b[k]), (name, k)


def test_fused_bottleneck_plan_same_outputs(monkeypatch):
    """fp16 plans run the Bottlenecks of 32-channel C3 blocks as one launch each (engine._Planner.c3 -> y5_bottleneck_fwd, with C3's
    cv1 half stored to its own buffer by the split store): same result as the conv + conv(+ residual) form, here on yolov5n whose
    layers 4 (two Bottlenecks: ping-pong buffers) and 17 (shortcut=False) qualify."""
    m = det_model("yolov5n", 0, True).half()
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=0)).half()
    monkeypatch.setenv("Y5_FUSED_BNECK", "0")
    plain = Engine(m, (2, 3, 64, 64), torch.float16, "cpu", want_raw=True, backend=EmuBackend())
    monkeypatch.setenv("Y5_FUSED_BNECK", "1")
    monkeypatch.setenv("Y5_FUSED_BNECK128", "force")   # (the c_ = 128 form is gated on its workgroup count: below the gate at this batch)
    fused = Engine(m, (2, 3, 64, 64), torch.float16, "cpu", want_raw=True, backend=EmuBackend())
    kinds = [o["op"] for o in fused.spec.ops]
    # 4.C3.m0 / m1, 17.C3.m0 (c_ = 32) and -- round 5 -- 8.C3.m0, 23.C3.m0 (c_ = 128: conv_h3b.h, here on 2 x 2 images)
    assert [o["x"].C for o in fused.spec.ops if o["op"] == "bneck"] == [32, 32, 128, 32, 128] and "bneck" not in [o["op"] for o in plain.spec.ops]
    assert sum(1 for o in fused.spec.ops if o.get("split_n")) == 4
    a, b = plain(x), fused(x)
    for k in a:  # same arithmetic, another fp32 summation order inside the 3x3 (two accumulators): a few fp16 ulps
        u, v = np.asarray(a[k]).astype(np.float32), np.asarray(b[k]).astype(np.float32)
        assert np.abs(u - v).max() <= 4e-3 * max(1.0, np.abs(u).max()), (k, np.abs(u - v).max())
        assert (u != v).mean() < 0.2, k


def test_virtual_upsample_concat_plan_same_outputs(monkeypatch):
    """fp16 plans read `nn.Upsample(2) -> Concat -> C3` (models/yolov5s.yaml:36-38,41-43) without the 2x replica: the C3's cv1+cv2 GEMM takes its first
    input channels from the low-resolution tensor (engine._Planner.run `virt`, conv configurations 88 / 89).  Same function as the plan that
    materialises the replica (another tile configuration, hence another fp32 summation order: a few fp16 ulps), and one replicated store less."""
    m = det_model("yolov5s", 0, True).half()
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=0)).half()
    monkeypatch.setenv("Y5_DISABLE", "virtual_up")
    plain = Engine(m, (2, 3, 64, 64), torch.float16, "cpu", want_raw=True, backend=EmuBackend())
    monkeypatch.setenv("Y5_DISABLE", "")
    virt = Engine(m, (2, 3, 64, 64), torch.float16, "cpu", want_raw=True, backend=EmuBackend())
    ups = [o["name"] for o in virt.spec.ops if o.get("op") == "conv" and o.get("x_up") is not None]
    assert ups == ["13.C3.cv1+cv2", "17.C3.cv1+cv2"] and not any(o.get("x_up") for o in plain.spec.ops if o.get("op") == "conv")
    rep = lambda e: [o["name"] for o in e.spec.ops if o.get("op") == "conv" and o.get("y2") is not None and not o.get("split_n")]  # noqa: E731
    assert rep(plain) == ["10.Conv", "14.Conv"] and rep(virt) == []
    a, b = plain(x), virt(x)
    for k in a:
        u, v = np.asarray(a[k]).astype(np.float32), np.asarray(b[k]).astype(np.float32)
        assert np.abs(u - v).max() <= 4e-3 * max(1.0, np.abs(u).max()), (k, np.abs(u - v).max())
