"""Kernel-placement experiment (GPU): the NMS of batch i on a HIGH-PRIORITY side stream beside the forward of batch i+1, against DetectPipeline
(everything on one stream).  Measurement only: the side-stream variant lets forward i+1 overwrite the engine's z buffer while NMS i may still read it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolov5_amd.detect_loop import DetectPipeline
from yolov5_amd.general import non_max_suppression

dev = torch.device("cuda:0")
model = bench.build_model("yolov5s", dev)
model.model[-1].export = True
x = torch.rand((64, 3, 640, 640)).half().to(dev)
bench.calibrate_head(model, x)
for _ in range(10): non_max_suppression(model(x)[0], 0.25, 0.45, max_det=1000)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
side = torch.cuda.Stream(dev, priority=-1)
main = torch.cuda.current_stream(dev)
pinned = [torch.empty((64,), dtype=torch.int32, pin_memory=True) for _ in range(3)]

def run_overlap(n):
    inflight = []
    got = 0
    for i in range(n):
        z = model(x)[0]
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            det, cnt = non_max_suppression(z, 0.25, 0.45, max_det=1000, padded=True)
            h = pinned[i % 3]; h.copy_(cnt, non_blocking=True)
            done = torch.cuda.Event(); done.record(side)
        inflight.append((det, h, done))
        if len(inflight) > 1:
            d, hh, e = inflight.pop(0); e.synchronize(); got += len(hh.tolist())
    while inflight:
        d, hh, e = inflight.pop(0); e.synchronize(); got += len(hh.tolist())
    return got

pipe = DetectPipeline(model, 0.25, 0.45, max_det=1000)
for rep in range(3):
    for _ in range(8): r = pipe.submit(x)
    pipe.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): r = pipe.submit(x)
    pipe.flush(); torch.cuda.synchronize()
    t_pipe = (time.perf_counter() - t0) / 50
    run_overlap(8); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_overlap(50); torch.cuda.synchronize()
    t_ov = (time.perf_counter() - t0) / 50
    print(f"one stream {t_pipe*1e3:.3f} ms/step ({64/t_pipe:.0f} img/s)   NMS on a high-priority side stream {t_ov*1e3:.3f} ms/step ({64/t_ov:.0f} img/s)")
