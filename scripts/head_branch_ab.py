#!/usr/bin/env python
"""A/B of the Detect-head side branch (Y5_HEAD_BRANCH): forward time of the headline configuration with the heads of the lower pyramid
levels on the plan's second stream vs everything on one stream, hipGraph replay and eager launches, same process and box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 4)


dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand((64, 3, 640, 640), device=dev).half()
res = {}
outs = {}
os.environ["Y5_TUNE_CACHE"] = "/tmp/y5_tune_ab.json"
for mode in ("0", "1", "0", "1"):
    os.environ["Y5_HEAD_BRANCH"] = mode
    m = bench.build_model("yolov5s", dev)
    with torch.no_grad():
        z = m(x)[0]
        eng = next(iter(m._engines.values()))
        g = timed(lambda: m(x))
        eng._use_graph = False
        e = timed(lambda: m(x))
        res.setdefault("branch" + mode, []).append({"graph_ms": g, "eager_ms": e})
        outs[mode] = m(x)[0].float().clone()
    del m, eng
    torch.cuda.empty_cache()
res["outputs_equal"] = bool(torch.equal(outs["0"], outs["1"]))
print(json.dumps(res))
