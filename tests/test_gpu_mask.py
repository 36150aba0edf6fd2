"""GPU parity of yolov5_amd.segment.process_mask (HIP y5_process_mask via the C-ABI) vs the reference golden masks and,
at segment/predict.py size (proto 32x160x160 -> 640x640, ~100 instances), vs the CPU oracle."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import detgen, yolo_oracle as yo

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask.npz"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("up,key", [(False, "noup"), (True, "up")])
def test_process_mask_vs_reference_golden(up, key, dev):
    from yolov5_amd.segment import process_mask

    protos = torch.from_numpy(detgen.uniform((32, 40, 40), -1.0, 1.0, name="protos", seed=13))
    coef = torch.from_numpy(detgen.uniform((7, 32), -1.0, 1.0, name="coef", seed=13))
    xy1 = detgen.uniform((7, 2), 0, 90, name="bx1", seed=13)
    wh = detgen.uniform((7, 2), 8, 70, name="bwh", seed=13)
    boxes = torch.from_numpy(np.concatenate((xy1, xy1 + wh), 1).astype(np.float32))
    m = process_mask(protos.to(dev), coef.to(dev), boxes.to(dev), (160, 160), upsample=up)
    assert m.dtype == torch.float32  # reference: masks.gt_(0.5) keeps the float dtype
    ref = np.unpackbits(G[f"m_{key}"])[: m.numel()].reshape(G[f"shape_{key}"]).astype(bool)
    assert (m.cpu().numpy().astype(bool) != ref).mean() < 1e-4
    mb = process_mask(protos.to(dev), coef.to(dev), boxes.to(dev), (160, 160), upsample=up, out_dtype=torch.bool)
    assert mb.dtype == torch.bool and torch.equal(mb, m.bool())


def test_process_mask_predict_size_on_nms_rows(dev):
    """cfg 5 shape: proto (32,160,160) fp16, 100 instances taken in place from NMS-style rows (n, 6+32), 640x640 output."""
    from yolov5_amd.segment import process_mask

    n = 100
    protos = detgen.uniform((32, 160, 160), -1.5, 1.5, name="P5", seed=5).astype(np.float16)
    det = np.zeros((n, 38), np.float32)
    xy1 = detgen.uniform((n, 2), 0, 500, name="b5", seed=5)
    det[:, :2] = xy1
    det[:, 2:4] = xy1 + detgen.uniform((n, 2), 10, 300, name="w5", seed=5)
    det[:, 6:] = detgen.uniform((n, 32), -1, 1, name="c5", seed=5)
    t0 = time.time()
    ref = yo.process_mask(torch.from_numpy(protos.astype(np.float32)), torch.from_numpy(det[:, 6:].copy()),
                          torch.from_numpy(det[:, :4].copy()), (640, 640), upsample=True).numpy().astype(bool)
    cpu_s = time.time() - t0
    d = torch.from_numpy(det).to(dev)
    P = torch.from_numpy(protos).to(dev)
    m = process_mask(P, d[:, 6:], d[:, :4], (640, 640), upsample=True, out_dtype=torch.bool)
    assert m.shape == (n, 640, 640)
    assert (m.cpu().numpy() != ref).mean() < 1e-4
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        process_mask(P, d[:, 6:], d[:, :4], (640, 640), upsample=True, out_dtype=torch.bool)
    torch.cuda.synchronize()
    gpu_ms = (time.time() - t0) / 20 * 1e3
    print(f"\n[process_mask] 100 instances 160^2 -> 640^2: {gpu_ms:.3f} ms on MI355X (uint8 out, {n * 640 * 640 / gpu_ms / 1e6:.1f} GB/s written); CPU oracle {cpu_s * 1e3:.0f} ms")


def test_process_mask_batch_equals_per_image_and_reports_write_rate(dev):
    """BASELINE config C5 shape (yolov5s-seg bs = 32: prototypes (32, 32, 160, 160) fp16, up to 300 detections per image taken in place from the padded NMS rows,
    640 x 640 masks): y5_process_mask_batch (ONE launch) == the per-image kernel pixel for pixel, in the reference's float32 and in uint8; image 3 has no
    detection.  Prints the write rate."""
    from yolov5_amd.segment import process_mask, process_mask_batch

    B, c, nmax = 32, 32, 300
    g = torch.Generator().manual_seed(3)
    P = (torch.randn((B, c, 160, 160), generator=g) * 0.8).half().to(dev)
    out = torch.zeros((B, nmax, 6 + c), dtype=torch.float32)
    cxy = torch.rand((B, nmax, 2), generator=g) * 640
    wh = torch.rand((B, nmax, 2), generator=g) ** 2 * 400 + 4
    out[..., :2] = (cxy - wh / 2).clamp(0, 640)
    out[..., 2:4] = (cxy + wh / 2).clamp(0, 640)
    out[..., 6:] = torch.randn((B, nmax, c), generator=g) * 0.5
    out = out.to(dev)
    counts = [nmax] * B
    counts[3], counts[7], counts[20] = 0, 1, 37
    dets = [out[i, :n] for i, n in enumerate(counts)]
    for dt in (torch.float32, torch.uint8):
        got = process_mask_batch(P, dets, (640, 640), upsample=True, out_dtype=dt)
        assert [tuple(m.shape) for m in got] == [(n, 640, 640) for n in counts]
        for i in (0, 3, 7, 20, 31):
            exp = process_mask(P[i], dets[i][:, 6:], dets[i][:, :4], (640, 640), upsample=True, out_dtype=dt)
            assert torch.equal(got[i], exp), i
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            process_mask_batch(P, dets, (640, 640), upsample=True, out_dtype=dt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        e0.record()
        for _ in range(2):
            [process_mask(P[i], d[:, 6:], d[:, :4], (640, 640), upsample=True, out_dtype=dt) for i, d in enumerate(dets) if len(d)]
        e1.record()
        torch.cuda.synchronize()
        ms_old = e0.elapsed_time(e1) / 2
        nbytes = sum(counts) * 640 * 640 * (4 if dt == torch.float32 else 1)
        print(f"\n[process_mask_batch] {sum(counts)} instances of 32 images, {dt}: one launch {ms:.3f} ms = {nbytes / ms / 1e9:.2f} TB/s written; "
              f"per-image launches {ms_old:.3f} ms = {nbytes / ms_old / 1e9:.2f} TB/s")
    # no-upsample form and the fallback for widths the batched kernel does not take
    got = process_mask_batch(P, dets, (640, 640), upsample=False)
    assert torch.equal(got[0], process_mask(P[0], dets[0][:, 6:], dets[0][:, :4], (640, 640), upsample=False))
    got = process_mask_batch(P[:2], dets[:2], (322, 322), upsample=True, out_dtype=torch.uint8)
    assert torch.equal(got[1], process_mask(P[1], dets[1][:, 6:], dets[1][:, :4], (322, 322), upsample=True, out_dtype=torch.uint8))
