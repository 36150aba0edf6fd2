"""A/B: sequential forward -> NMS steps vs DetectPipeline (batch i+1 queued before the host waits for batch i's counts), yolov5s bs=64."""
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
def seq():
    return non_max_suppression(model(x)[0], 0.25, 0.45, max_det=1000)
for _ in range(10): seq()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50): seq()
    torch.cuda.synchronize()
    t_seq = (time.perf_counter() - t0) / 50
    pipe = DetectPipeline(model, 0.25, 0.45, max_det=1000)
    for _ in range(8): r = pipe.submit(x)
    pipe.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): pipe.submit(x)
    pipe.flush()
    torch.cuda.synchronize()
    t_pipe = (time.perf_counter() - t0) / 50
    print(f"sequential {t_seq*1e3:.3f} ms/step ({64/t_seq:.0f} img/s)   pipeline {t_pipe*1e3:.3f} ms/step ({64/t_pipe:.0f} img/s)")
