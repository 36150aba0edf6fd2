"""Host-side phase times of DetectPipeline.submit (debug aid)."""
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
for _ in range(10): non_max_suppression(model(x)[0], 0.25, 0.45, max_det=1000)
torch.cuda.synchronize()
print("cpus", os.cpu_count(), "threads", torch.get_num_threads(), flush=True)
pinned = [torch.empty((64,), dtype=torch.int32, pin_memory=True) for _ in range(2)]
for mode in ("pinned_copy", "device_clone"):
    prev = None
    acc = [0.0] * 6
    n = 0
    for i in range(60):
        t0 = time.perf_counter()
        z = model(x)[0]
        t1 = time.perf_counter()
        det, cnt = non_max_suppression(z, 0.25, 0.45, max_det=1000, padded=True)
        t2 = time.perf_counter()
        if mode == "pinned_copy":
            h = pinned[i & 1]; h.copy_(cnt, non_blocking=True)
        else:
            h = cnt.clone()
        t3 = time.perf_counter()
        ev = torch.cuda.Event(); ev.record()
        t4 = time.perf_counter()
        if prev is not None:
            prev[2].synchronize()
            t5 = time.perf_counter()
            c = prev[1].tolist()
            t6 = time.perf_counter()
        else:
            t5 = t6 = t4
        prev = (det, h, ev)
        if i >= 10:
            for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))): acc[k] += b - a
            n += 1
    torch.cuda.synchronize()
    print(mode, "host us per step: model(x) %.0f  nms() %.0f  count copy %.0f  event %.0f  wait %.0f  tolist %.0f   total %.0f" % tuple([v / n * 1e6 for v in acc] + [sum(acc) / n * 1e6]), flush=True)
