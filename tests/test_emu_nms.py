"""CPU (no GPU): NMS / decode / pool / layout kernels run on the HIP emulator against the golden vectors and the
oracle -- bit-exact selection order (integer/index work), fp tolerance for decode."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo
from oracle.make_golden import NMS_CASES, nms_case_pred
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib

G = os.path.join(os.path.dirname(__file__), "golden")


def run_nms(lib, pred, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300,
            nm=0, max_nms=30000, dtype=np.float32):
    bs, n, no = pred.shape
    p = aligned(pred.shape, dtype); p[...] = pred.astype(dtype)
    flags = (_lib.NMS_MULTI_LABEL if multi_label else 0) | (_lib.NMS_AGNOSTIC if agnostic else 0)
    wsb = lib.y5_nms_workspace_bytes(bs, n, no, nm, flags, max_nms)
    ws = aligned((wsb,), np.uint8)
    out = aligned((bs, max_det, 6 + nm), np.float32, -1.0)
    cnt = aligned((bs,), np.int32, -1)
    cls = None
    if classes is not None:
        cls = aligned((len(classes),), np.int32); cls[...] = classes
    rc = lib.y5_nms_batched(ptr(p), _lib.Y5_F16 if dtype == np.float16 else _lib.Y5_F32, bs, n, no, nm, conf_thres, iou_thres,
                            max_det, max_nms, 7680.0, flags, ptr(cls), 0 if cls is None else len(classes), ptr(out), ptr(cnt),
                            ptr(ws), wsb, None)
    assert rc == 0, lib.y5_last_error()
    return [out[i, :cnt[i]].copy() for i in range(bs)]


@pytest.mark.parametrize("name", list(NMS_CASES))
def test_nms_emulated_bit_exact_vs_golden(name):
    g = np.load(os.path.join(G, "nms.npz"))
    kw, nkw = NMS_CASES[name]
    res = run_nms(emu(), nms_case_pred(name), **nkw)
    for i, r in enumerate(res):
        ref = g[f"{name}_{i}"]
        assert r.shape == ref.shape, (name, i, r.shape, ref.shape)
        assert np.array_equal(r, ref), (name, i)


def test_nms_fp16_input_equals_oracle_on_upcast():
    """Contract for half predictions: identical to the fp32 path on pred.float() (NMS dtype note)."""
    p = detgen.synth_predictions(2, 1500, 85, obj_pow=4, seed=21).astype(np.float16)
    res = run_nms(emu(), p, conf_thres=0.25, iou_thres=0.45, max_det=300, dtype=np.float16)
    ref = yo.non_max_suppression(p.astype(np.float32), 0.25, 0.45, max_det=300)
    for r, o in zip(res, ref):
        assert np.array_equal(r, o)


def test_nms_large_candidate_set_global_sort_path():
    """> 8192 candidates per image exercises the global-memory bitonic path and the max_nms truncation."""
    p = detgen.synth_predictions(1, 12000, 9, obj_pow=1, seed=22)
    kw = dict(conf_thres=0.01, iou_thres=0.5, multi_label=True, max_det=100, max_nms=9000)
    res = run_nms(emu(), p, **kw)
    ref = yo.non_max_suppression(p, **kw)
    assert len(ref[0]) == 100 and np.array_equal(res[0], ref[0])


def test_nms_pruning_stage_short_lists_ties_and_overflow():
    """K1c (csrc/nms_kernels.h): lists longer than max_nms are cut to the confidence bins that can hold the max_nms best keys before the sort.
    One call, three images: a long random list (pruned), a list shorter than max_nms (left alone), and a list whose candidates all share ONE
    confidence (every key in the same bin: the compaction buffer overflows and the image falls back to sorting everything) -- each bit-identical
    to the oracle, whose order among equal confidences is the candidate index (the contract of the keys' low word)."""
    p = detgen.synth_predictions(3, 3000, 9, obj_pow=1, seed=23)
    p[1, 150:, 4] = 0.0                                   # image 1: few rows survive the objectness threshold
    p[2, :, 4] = 0.5; p[2, :, 5:] = 0.5                   # image 2: every (row, class) has confidence 0.25
    kw = dict(conf_thres=0.01, iou_thres=0.5, multi_label=True, max_det=60, max_nms=500)
    res = run_nms(emu(), p, **kw)
    ref = yo.non_max_suppression(p, **kw)
    for i in range(3):
        assert len(ref[i]) > 0 and np.array_equal(res[i], ref[i]), i
    # exactly max_nms candidates at the boundary, and max_nms larger than the list
    for mx in (499, 501, 11990, 40000):
        kw["max_nms"] = mx
        res = run_nms(emu(), p[:1], **kw)
        ref = yo.non_max_suppression(p[:1], **kw)
        assert np.array_equal(res[0], ref[0]), mx


def test_detect_decode_emulated():
    lib = emu()
    B, ny, nx, na, no = 2, 5, 7, 3, 85
    logits = detgen.uniform((B, ny, nx, na * no), -4, 4, name="lg", seed=3)
    ld = 256
    lg = aligned((B, ny, nx, ld), np.float32, 9.0); lg[..., : na * no] = logits
    anchors = np.array([[1.25, 1.625], [2.0, 3.75], [4.125, 2.875]], np.float32)
    stride = 8.0
    apx = (anchors * stride).astype(np.float32).reshape(-1)
    nrows = na * ny * nx + 11
    z = aligned((B, nrows, no), np.float32, -1.0)
    raw = aligned((B, na, ny, nx, no), np.float32, -1.0)
    rc = lib.y5_detect_decode(ptr(lg), _lib.Y5_F32, B, ny, nx, na, no, 0, ld, stride, apx.ctypes.data_as(C.POINTER(C.c_float)),
                              ptr(z), _lib.Y5_F32, nrows, 11, ptr(raw), None)
    assert rc == 0, lib.y5_last_error()
    x = torch.from_numpy(logits).view(B, ny, nx, na, no).permute(0, 3, 1, 2, 4).contiguous()
    assert np.array_equal(raw, x.numpy())
    grid, ag = yo.make_grid(nx, ny, torch.from_numpy(anchors), stride)
    s = x.sigmoid()
    ref = torch.cat(((s[..., :2] * 2 + grid) * stride, (s[..., 2:4] * 2) ** 2 * ag, s[..., 4:]), 4).view(B, -1, no).numpy()
    np.testing.assert_allclose(z[:, 11:], ref, rtol=1e-5, atol=1e-5)
    assert np.all(z[:, :11] == -1.0)


@pytest.mark.parametrize("ny,nx,no,nm,row_off", [(8, 8, 85, 0, 0), (4, 16, 85, 0, 192), (8, 8, 117, 32, 0), (8, 4, 13, 0, 64)])
def test_detect_decode_fp16_paths_agree(ny, nx, no, nm, row_off):
    """fp16 decode: the z-only path (box outputs pre-computed in the LDS tile, sigmoid-only element loop) against the general
    branch-free path (taken when the raw tensor is requested) and against the unaligned fallbacks: bit-identical z."""
    lib = emu()
    B, na, ld = 2, 3, 352 if no > 85 else 256
    lg = aligned((B, ny, nx, ld), np.float16, 1.0)
    lg[..., : na * no] = detgen.uniform((B, ny, nx, na * no), -5, 5, name="lgh", seed=ny * nx + no).astype(np.float16)
    apx = (C.c_float * 6)(10, 13, 16, 30, 33, 23)
    nrows = row_off + na * ny * nx
    outs = []
    for with_raw in (False, True):
        z = aligned((B, nrows, no), np.float16, -1.0)
        raw = aligned((B, na, ny, nx, no), np.float16, -1.0)
        rc = lib.y5_detect_decode(ptr(lg), _lib.Y5_F16, B, ny, nx, na, no, nm, ld, 8.0, apx, ptr(z), _lib.Y5_F16, nrows, row_off,
                                  ptr(raw) if with_raw else None, None)
        assert rc == 0, lib.y5_last_error()
        outs.append(z.copy())
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))
    # and against the float formulas (models/yolo.py:104-113)
    x = torch.from_numpy(lg[..., : na * no].astype(np.float32)).view(B, ny, nx, na, no).permute(0, 3, 1, 2, 4)
    grid, ag = yo.make_grid(nx, ny, torch.tensor([[10 / 8, 13 / 8], [16 / 8, 30 / 8], [33 / 8, 23 / 8]]), 8.0)
    sg = x.sigmoid()
    nact = no - nm
    ref = torch.cat(((sg[..., :2] * 2 + grid) * 8.0, (sg[..., 2:4] * 2) ** 2 * ag, sg[..., 4:nact], x[..., nact:]), 4).reshape(B, -1, no).numpy()
    np.testing.assert_allclose(outs[0][:, row_off:].astype(np.float32), ref, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("dt,Cc,H,W", [("f16", 16, 6, 5), ("f32", 16, 6, 5), ("f16", 128, 6, 5), ("f32", 64, 6, 5),   # 128 x f16 / 64 x f32: 128-byte channel groups
                                       ("f16", 32, 20, 20),    # the yolov5 P5 plane at 640^2: several vectors per thread, windows precomputed per thread
                                       ("f16", 16, 44, 48),    # more than 8 vectors per thread: generic index path, separable
                                       ("f16", 16, 60, 56)])   # plane too large for the third LDS plane: direct k x k window
def test_sppf_pool_emulated(dt, Cc, H, W):
    lib = emu()
    npdt = np.float16 if dt == "f16" else np.float32
    B = 2 if H * W < 1000 else 1
    x = detgen.uniform((B, Cc, H, W), -2, 2, name="pool").astype(npdt)
    buf = aligned((B, H, W, 4 * Cc), npdt, 0.0)
    buf[..., :Cc] = x.transpose(0, 2, 3, 1)
    rc = lib.y5_sppf_pool(ptr(buf), _lib.Y5_F16 if dt == "f16" else _lib.Y5_F32, B, H, W, Cc, 4 * Cc, 5, None)
    assert rc == 0, lib.y5_last_error()
    t = torch.from_numpy(x.astype(np.float32))
    for i in range(1, 4):
        t = torch.nn.functional.max_pool2d(t, 5, 1, 2)
        assert np.array_equal(buf[..., i * Cc:(i + 1) * Cc].astype(np.float32), t.permute(0, 2, 3, 1).numpy())


def test_sppf_pool_bwd_exact_and_nonfinite_emulated():
    """The fixed-point scatter of y5_sppf_pool_bwd is EXACT: gradients spanning fp16's whole range (subnormals 6e-8 ... 6e4, both signs) sum to the correctly
    rounded value of the float64 sum (one fp32 rounding per pass, one fp16 rounding at the end -- never further than 1.01 fp16 ulp from the float64 result);
    an Inf or NaN in ANY incoming gradient slice comes out non-finite (the loss scaler has to see the overflow), in the gather fallback as in the scatter form."""
    lib = emu()
    B, H, W, Cc = 1, 6, 7, 8
    rng = np.random.default_rng(5)
    x = torch.from_numpy((rng.integers(0, 6, (B, Cc, H, W)) / 2.0).astype(np.float32)).double().requires_grad_(True)
    y1 = torch.nn.functional.max_pool2d(x, 5, 1, 2)
    y2 = torch.nn.functional.max_pool2d(y1, 5, 1, 2)
    y3 = torch.nn.functional.max_pool2d(y2, 5, 1, 2)
    mag = 2.0 ** rng.integers(-24, 14, (4, B, Cc, H, W))
    gs = [torch.from_numpy((mag[i] * rng.choice([-1.0, 1.0, 1.5], (B, Cc, H, W))).astype(np.float16).astype(np.float64)) for i in range(4)]
    (x * gs[0] + y1 * gs[1] + y2 * gs[2] + y3 * gs[3]).sum().backward()
    act = aligned((B, H, W, 4 * Cc), np.float16, 0.0)
    grad = aligned((B, H, W, 4 * Cc), np.float16, 0.0)
    for i, t in enumerate((x, y1, y2, y3)):
        act[..., i * Cc:(i + 1) * Cc] = t.detach().permute(0, 2, 3, 1).numpy()
        grad[..., i * Cc:(i + 1) * Cc] = gs[i].permute(0, 2, 3, 1).numpy()
    g0 = grad.copy()
    assert lib.y5_sppf_pool_bwd(ptr(act), ptr(grad), B, H, W, Cc, 4 * Cc, 4 * Cc, 5, None) == 0, lib.y5_last_error()
    ref = x.grad.permute(0, 2, 3, 1).numpy()
    got = grad[..., :Cc].astype(np.float64)
    fin = np.abs(ref) < 65000
    ulp = np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10), 2.0 ** -24)
    assert np.all(np.abs(got - ref)[fin] <= 1.01 * ulp[fin]), float(np.max((np.abs(got - ref) / ulp)[fin]))
    for bad in (np.inf, np.nan):
        grad[...] = g0
        grad[0, 2, 3, 2 * Cc + 1] = bad        # one element of the d/dy2 slice
        assert lib.y5_sppf_pool_bwd(ptr(act), ptr(grad), B, H, W, Cc, 4 * Cc, 4 * Cc, 5, None) == 0
        assert not np.isfinite(grad[..., :Cc].astype(np.float32)).all()   # (NaN from the scatter form, Inf / NaN from the gather form)


def test_sppf_pool_bwd_gather_form_emulated():
    """The gather fallback of y5_sppf_pool_bwd (planes too large for the fixed-point scatter; forced here with Y5_SPPF_BWD_GATHER=1, which the library
    latches per process)."""
    import os
    import subprocess
    import sys

    code = "import tests.test_emu_nms as t; [t.test_sppf_pool_bwd_emulated(c) for c in (8, 32)]; t.test_sppf_pool_bwd_exact_and_nonfinite_emulated()"
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, Y5_SPPF_BWD_GATHER="1"), capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("Cc", [8, 32, 64])   # 32 / 64: four 16-byte groups per workgroup
def test_sppf_pool_bwd_emulated(Cc):
    """y5_sppf_pool_bwd against torch autograd through three chained max_pool2d(5, 1, 2) (models/common.py:338-340)."""
    lib = emu()
    B, H, W = 2, 7, 6
    x = torch.from_numpy(detgen.uniform((B, Cc, H, W), -2, 2, name="pbx")).half().float().requires_grad_(True)
    y1 = torch.nn.functional.max_pool2d(x, 5, 1, 2)
    y2 = torch.nn.functional.max_pool2d(y1, 5, 1, 2)
    y3 = torch.nn.functional.max_pool2d(y2, 5, 1, 2)
    gs = [torch.from_numpy(detgen.uniform((B, Cc, H, W), -1, 1, name=f"pbg{i}")).half().float() for i in range(4)]
    (x * gs[0] + y1 * gs[1] + y2 * gs[2] + y3 * gs[3]).sum().backward()
    act = aligned((B, H, W, 4 * Cc), np.float16, 0.0)
    grad = aligned((B, H, W, 4 * Cc), np.float16, 0.0)
    for i, t in enumerate((x, y1, y2, y3)):
        act[..., i * Cc:(i + 1) * Cc] = t.detach().permute(0, 2, 3, 1).numpy()
        grad[..., i * Cc:(i + 1) * Cc] = gs[i].permute(0, 2, 3, 1).numpy()
    rc = lib.y5_sppf_pool_bwd(ptr(act), ptr(grad), B, H, W, Cc, 4 * Cc, 4 * Cc, 5, None)
    assert rc == 0, lib.y5_last_error()
    np.testing.assert_allclose(grad[..., :Cc].astype(np.float32), x.grad.permute(0, 2, 3, 1).numpy(), rtol=2e-3, atol=2e-3)


def test_layout_and_copy_kernels_emulated():
    lib = emu()
    B, Cc, H, W = 2, 3, 5, 6
    img = (detgen.uniform((B, Cc, H, W), 0, 255.99, name="u8")).astype(np.uint8)
    src = aligned(img.shape, np.uint8); src[...] = img
    dst = aligned((B, H, W, 4), np.float16, 5.0)
    assert lib.y5_nchw_to_nhwc(ptr(src), _lib.Y5_U8, ptr(dst), _lib.Y5_F16, B, Cc, H, W, 4, 1.0 / 255.0, None) == 0
    ref = (img.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float16).transpose(0, 2, 3, 1)
    assert np.array_equal(dst[..., :3], ref) and np.all(dst[..., 3] == 0)
    # nhwc slice -> nchw
    a = aligned((B, H, W, 16), np.float32); a[...] = detgen.uniform(a.shape, -1, 1, name="a")
    o = aligned((B, 8, H, W), np.float32)
    assert lib.y5_nhwc_to_nchw(ptr(a[..., 4:]), _lib.Y5_F32, ptr(o), B, 8, H, W, 16, None) == 0
    assert np.array_equal(o, a[..., 4:12].transpose(0, 3, 1, 2))
    # upsample into a slice, copy slice
    s = aligned((B, H, W, 8), np.float16); s[...] = detgen.uniform(s.shape, -1, 1, name="s")
    d = aligned((B, 2 * H, 2 * W, 24), np.float16, 3.0)
    assert lib.y5_upsample2x(ptr(s), _lib.Y5_F16, ptr(d[..., 8:]), B, H, W, 8, 8, 24, None) == 0
    assert np.array_equal(d[..., 8:16], np.repeat(np.repeat(s, 2, 1), 2, 2)) and np.all(d[..., :8] == 3) and np.all(d[..., 16:] == 3)
    e = aligned((B, H, W, 24), np.float16, 1.0)
    assert lib.y5_copy_slice(ptr(s), _lib.Y5_F16, ptr(e[..., 16:]), B * H * W, 8, 8, 24, None) == 0
    assert np.array_equal(e[..., 16:], s) and np.all(e[..., :16] == 1)


# ---- objectness hint plane (y5_detect_decode_hint -> y5_nms_batched_hint) ------------------------------------------------------------
def _run_nms_hint(lib, pred, hint, dtype, **kw):
    bs, n, no = pred.shape
    p = aligned(pred.shape, dtype); p[...] = pred.astype(dtype)
    h = aligned((bs, n), dtype); h[...] = hint.astype(dtype)
    max_det, max_nms = kw.get("max_det", 300), 30000
    flags = _lib.NMS_MULTI_LABEL if kw.get("multi_label") else 0
    wsb = lib.y5_nms_workspace_bytes(bs, n, no, 0, flags, max_nms)
    ws = aligned((wsb,), np.uint8)
    out = aligned((bs, max_det, 6), np.float32, -1.0)
    cnt = aligned((bs,), np.int32, -1)
    rc = lib.y5_nms_batched_hint(ptr(p), _lib.Y5_F16 if dtype == np.float16 else _lib.Y5_F32, bs, n, no, 0, kw.get("conf_thres", 0.25), kw.get("iou_thres", 0.45),
                                 max_det, max_nms, 7680.0, flags, None, 0, ptr(out), ptr(cnt), ptr(ws), wsb, ptr(h), None)
    assert rc == 0, lib.y5_last_error()
    return [out[i, :cnt[i]].copy() for i in range(bs)]


@pytest.mark.parametrize("dtype,multi", [(np.float16, False), (np.float32, False), (np.float16, True)])
def test_nms_with_objectness_hint_is_the_plain_result(dtype, multi):
    """The plane is a hint: rows it cannot exclude with the margin are decided on `pred` itself.  Exact copy of pred[..., 4], a copy off by
    one / two fp16 ulps in either direction, and a plane that says 1.0 everywhere all give the detections of the plain filter."""
    lib = emu()
    p = detgen.synth_predictions(2, 1500, 85, obj_pow=4, seed=31).astype(dtype)
    kw = dict(conf_thres=0.25, iou_thres=0.45, max_det=300, multi_label=multi)
    ref = run_nms(lib, p, dtype=dtype, **kw)
    obj = p[..., 4].astype(np.float16)
    ulp = lambda a, k: (a.view(np.int16) + np.int16(k)).view(np.float16)  # noqa: E731
    for hint in (obj, ulp(obj.copy(), 1), ulp(obj.copy(), -2), np.ones_like(obj)):
        res = _run_nms_hint(lib, p, hint.astype(np.float32), dtype, **kw)
        for r, o in zip(res, ref):
            assert np.array_equal(r, o)


def test_decode_hint_plane_equals_z_objectness():
    """y5_detect_decode_hint: the plane holds z[..., 4] of every row it decodes (same formula as the path that writes z), untouched elsewhere."""
    lib = emu()
    B, ny, nx, na, no, ld, row_off = 2, 8, 8, 3, 85, 256, 64
    lg = aligned((B, ny, nx, ld), np.float16, 1.0)
    lg[..., : na * no] = detgen.uniform((B, ny, nx, na * no), -5, 5, name="lgh2", seed=5).astype(np.float16)
    apx = (C.c_float * 6)(10, 13, 16, 30, 33, 23)
    nrows = row_off + na * ny * nx + 8
    for with_raw in (False, True):
        z = aligned((B, nrows, no), np.float16, -1.0)
        raw = aligned((B, na, ny, nx, no), np.float16, -1.0)
        hint = aligned((B, nrows), np.float16, -7.0)
        rc = lib.y5_detect_decode_hint(ptr(lg), _lib.Y5_F16, B, ny, nx, na, no, 0, ld, 8.0, apx, ptr(z), _lib.Y5_F16, nrows, row_off,
                                       ptr(raw) if with_raw else None, ptr(hint), None)
        assert rc == 0, lib.y5_last_error()
        sl = slice(row_off, row_off + na * ny * nx)
        assert np.array_equal(hint[:, sl].view(np.uint16), z[:, sl, 4].view(np.uint16))
        assert np.all(hint[:, :row_off] == -7.0) and np.all(hint[:, row_off + na * ny * nx:] == -7.0)
