"""GPU: NMS kernel times with and without the objectness plane (run under rocprofv3 --kernel-trace --stats to see the filter kernel alone)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolov5_amd.general import non_max_suppression

dev = torch.device("cuda:0")
model = bench.build_model("yolov5s", dev)
model.model[-1].export = True
x = torch.rand((64, 3, 640, 640)).half().to(dev)
bench.calibrate_head(model, x)
z = model(x)[0]
zc = z.clone()
for name, t in (("hint", z), ("plain", zc)):
    for _ in range(5):
        non_max_suppression(t, 0.25, 0.45, max_det=1000, padded=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        non_max_suppression(t, 0.25, 0.45, max_det=1000, padded=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per 64-image NMS (padded, no host sync)")
