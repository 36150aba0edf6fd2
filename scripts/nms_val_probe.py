"""NMS with val.py's thresholds (conf 0.001, iou 0.6, multi_label, max_det 300) on the dense synthetic distribution of bench.nms_distributions: per-kernel
breakdown under `rocprofv3 --kernel-trace --stats` and the event-timed figure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolov5_amd.general import non_max_suppression

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
p = torch.rand((64, 25200, 85), generator=g)
p[..., 0:2] *= 640.0
p[..., 2:4] = 4.0 + 100.0 * p[..., 2:4]
p[..., 4] = p[..., 4] ** 8
p = p.half().to(dev)
kw = dict(conf_thres=0.001, iou_thres=0.6, max_det=300, multi_label=True)
for _ in range(2):
    non_max_suppression(p, padded=True, **kw)
ms = bench.event_times(lambda: non_max_suppression(p, padded=True, **kw), 5, dev)
print(f"val-mode NMS: {bench._pct(ms, .5) * 1e3 / 64:.1f} us/img device side")
