import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import yolov5_amd.detect_loop as dl
from yolov5_amd.general import non_max_suppression
dev = torch.device("cuda:0")
model = bench.build_model("yolov5s", dev)
model.model[-1].export = True
x = torch.rand((64, 3, 640, 640)).half().to(dev)
bench.calibrate_head(model, x)
def seq(): return non_max_suppression(model(x)[0], 0.25, 0.45, max_det=1000)
for _ in range(10): seq()
T = {"model": 0.0, "nms": 0.0, "wait": 0.0, "n": 0}
orig_nms = dl.non_max_suppression
def timed_nms(*a, **k):
    t0 = time.perf_counter(); r = orig_nms(*a, **k); T["nms"] += time.perf_counter() - t0; return r
dl.non_max_suppression = timed_nms
class M:
    def __call__(self, x):
        t0 = time.perf_counter(); r = model(x); T["model"] += time.perf_counter() - t0; return r
orig_collect = dl.DetectPipeline._collect
def timed_collect(prev):
    t0 = time.perf_counter(); r = orig_collect(prev); T["wait"] += time.perf_counter() - t0; T["n"] += 1; return r
dl.DetectPipeline._collect = staticmethod(timed_collect)
pipe = dl.DetectPipeline(M(), 0.25, 0.45, max_det=1000)
def measure(tag, with_seq):
    if with_seq:
        for _ in range(45): seq()
    torch.cuda.synchronize()
    for _ in range(8): r = pipe.submit(x)
    pipe.flush(); torch.cuda.synchronize()
    for k in T: T[k] = 0
    t0 = time.perf_counter()
    for _ in range(40): r = pipe.submit(x)
    pipe.flush(); torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 40
    print(f"{tag}: {tp*1e3:.3f} ms/step; host us/step: model {T['model']/40*1e6:.0f} nms {T['nms']/40*1e6:.0f} collect {T['wait']/40*1e6:.0f}", flush=True)
measure("no seq before", False)
measure("no seq before (2)", False)
measure("45 seq() before", True)
measure("no seq before (3)", False)
