"""Is the forward time stable under sustained load on this box?  400 back-to-back forwards, per-forward HIP-event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
x = torch.rand((64, 3, 640, 640), device=dev).half()
m = bench.build_model("yolov5s", dev)
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    time.sleep(2.0)
    n = 400
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        m(x)
        ev[i + 1].record()
    torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for a, b in ((0, 5), (5, 20), (20, 50), (50, 100), (100, 200), (200, 400)):
    print(f"forwards {a:3d}-{b:3d}: mean {sum(t[a:b]) / (b - a):.3f} ms  min {min(t[a:b]):.3f}  max {max(t[a:b]):.3f}")
