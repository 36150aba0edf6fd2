"""Regression probe (GPU): step time of the plain loop and of DetectPipeline in one process, before / after other streams (a side stream, a
high-priority stream, an RCCL process group) exist.  DetectPipeline must stay below the plain loop in every line; 4.1 ms against 2.7 ms was the signature
of the per-forward graph re-capture fixed in round 2."""
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
def seq(): return non_max_suppression(model(x)[0], 0.25, 0.45, max_det=1000)
for _ in range(10): seq()
pipe = DetectPipeline(model, 0.25, 0.45, max_det=1000)

def measure(tag):
    for _ in range(5): seq()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): seq()
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 40
    for _ in range(8): r = pipe.submit(x)
    pipe.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): r = pipe.submit(x)
    pipe.flush(); torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 40
    print(f"{tag:48s} sequential {ts*1e3:.3f} ms   DetectPipeline {tp*1e3:.3f} ms", flush=True)

measure("no other stream")
s1 = torch.cuda.Stream(dev)
measure("+ an idle normal-priority stream")
with torch.cuda.stream(s1):
    y = torch.zeros(1024, device=dev) + 1
torch.cuda.synchronize()
measure("+ after work on it")
s2 = torch.cuda.Stream(dev, priority=-1)
measure("+ an idle high-priority stream")
with torch.cuda.stream(s2):
    y = torch.zeros(1024, device=dev) + 1
torch.cuda.synchronize()
measure("+ after work on the high-priority stream")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
measure("+ RCCL process group (world 1) used once")
