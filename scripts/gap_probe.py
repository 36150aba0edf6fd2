#!/usr/bin/env python
"""Why do some boxes run the same forward in 3.3 ms and others in 2.9 ms with identical per-kernel times?  Prints, for the
headline configuration: hipGraph replay vs eager launches vs the sum of the per-op times, the host time to enqueue one eager
forward, the host CPU model and the GPU clocks / power rocm-smi reports right after a sustained loop."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_enq / n * 1e3


dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand((64, 3, 640, 640), device=dev).half()
m = bench.build_model("yolov5s", dev)
with torch.no_grad():
    m(x)
    eng = next(iter(m._engines.values()))
    res = {"hipgraph_captured": bool(eng._graph)}
    res["graph_ms"], res["graph_enqueue_ms"] = timed(lambda: m(x))
    eng._use_graph = False
    res["eager_ms"], res["eager_enqueue_ms"] = timed(lambda: m(x))
    eng._use_graph = True
    ops = eng.time_ops(iters=10)
    res["sum_of_ops_ms"] = sum(ms for _, ms in ops)
    res["n_ops"] = len(ops)
    # sustained: 300 forwards back to back, then the clocks
    t0 = time.perf_counter()
    for _ in range(300):
        m(x)
    torch.cuda.synchronize()
    res["sustained_ms"] = (time.perf_counter() - t0) / 300 * 1e3
try:
    smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
    res["rocm_smi"] = json.loads(smi)
except Exception as e:  # noqa: BLE001
    res["rocm_smi"] = str(e)
try:
    with open("/proc/cpuinfo") as f:
        models = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
    res["cpu"] = {"model": models[0] if models else "?", "threads": len(models)}
except OSError:
    pass
res = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}
print(json.dumps(res))
