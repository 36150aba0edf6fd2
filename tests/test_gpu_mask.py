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
