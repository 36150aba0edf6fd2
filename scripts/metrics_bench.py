"""Validation matching at val.py's default batch (bs=32... here 64, max_det=300, ~7 labels per image, 10 IoU thresholds):
one y5_val_match launch behind the padded NMS result vs the CPU oracle's per-image loop (the reference's structure:
scale_boxes x2 + process_batch per image, val.py:282-307).  Prints one JSON line."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import detgen, yolo_oracle as yo  # noqa: E402  (checker + CPU baseline only)
from yolov5_amd.metrics import match_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    bs, max_det, nc, per = 64, 300, 80, 7
    rng = np.random.default_rng(0)
    t = detgen.synth_targets(bs, per, nc, seed=3)
    t[:, 2:] *= np.float32(640)
    det = np.zeros((bs, max_det, 6), np.float32)
    for si in range(bs):
        lab = t[t[:, 0] == si]
        pick = rng.integers(0, per, max_det)
        c = lab[pick, 2:4] + rng.normal(0, 4, (max_det, 2))
        wh = lab[pick, 4:6] * rng.uniform(0.8, 1.2, (max_det, 2))
        det[si, :, 0:2], det[si, :, 2:4] = c - wh / 2, c + wh / 2
        det[si, :, 4] = np.sort(rng.uniform(0, 1, max_det))[::-1]
        det[si, :, 5] = np.where(rng.uniform(0, 1, max_det) < 0.7, lab[pick, 1], rng.integers(0, nc, max_det))
    cnt = np.full((bs,), max_det, np.int32)
    shapes = [((480, 640), ((0.8, 0.8), (0.0, 64.0)))] * bs
    iouv = torch.linspace(0.5, 0.95, 10, device=dev)
    out, counts, targets = torch.from_numpy(det).to(dev), torch.from_numpy(cnt).to(dev), torch.from_numpy(t).to(dev)
    correct = match_batch(out, counts, targets, shapes, iouv)
    torch.cuda.synchronize()
    # parity on this exact input
    t0 = time.perf_counter()
    ref = [yo.val_match_image(det[si], t[t[:, 0] == si, 1:], (640, 640), shapes[si][0], shapes[si][1], iouv.cpu().numpy())[0] for si in range(bs)]
    cpu_ms = (time.perf_counter() - t0) * 1e3
    assert np.array_equal(correct.cpu().numpy().astype(bool), np.stack(ref))
    # device time of the launch alone (scale table resident) and of the host call
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        match_batch(out, counts, targets, shapes, iouv)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        match_batch(out, counts, targets, shapes, iouv)
    e1.record()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / n
    print(json.dumps({"op": "val_match", "bs": bs, "max_det": max_det, "labels": int(t.shape[0]), "niou": 10,
                      "gpu_ms_per_batch": round(e0.elapsed_time(e1) / n, 4), "host_call_ms": round(wall_ms, 4),
                      "cpu_oracle_ms_per_batch": round(cpu_ms, 2), "true_positives@0.5": int(correct[..., 0].sum())}))


if __name__ == "__main__":
    main()
