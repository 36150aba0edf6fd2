"""NMS alone on a bench-like prediction tensor (yolov5s bs=64 forward with the calibrated head): for rocprofv3 kernel stats and host-side timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolov5_amd.general import non_max_suppression

dev = torch.device("cuda:0")
model = bench.build_model("yolov5s", dev)
model.model[-1].export = True
x = torch.rand((64, 3, 640, 640), generator=torch.Generator().manual_seed(0)).half().to(dev)
bench.calibrate_head(model, x)
z = model(x)[0]
for _ in range(5):
    out = non_max_suppression(z, 0.25, 0.45, max_det=1000)
torch.cuda.synchronize()
ms = bench.event_times(lambda: non_max_suppression(z, 0.25, 0.45, max_det=1000), 50, dev)
ms2 = bench.event_times(lambda: non_max_suppression(z, 0.25, 0.45, max_det=1000, padded=True), 50, dev)
obj = z[..., 4].float()
cand = ((z[..., 5:].float() * obj[..., None]).max(-1).values > 0.25) & (obj > 0.25)
print(f"nms median {bench._pct(ms, .5)*1e3:.1f} us/batch (padded, no host sync: {bench._pct(ms2, .5)*1e3:.1f}); candidates/img mean {cand.sum(1).float().mean().item():.0f} "
      f"max {cand.sum(1).max().item()}; kept/img {sum(len(o) for o in out)/64:.0f}")
