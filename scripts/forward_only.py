#!/usr/bin/env python
"""Exactly N forwards of the headline configuration (yolov5s, 64 x 3x640x640 fp16), nothing else on the GPU -- the target of
the PMC passes of scripts/pmc_forward.sh (run once with Y5_TUNE_CACHE set to fill the tile-choice cache first)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand((64, 3, 640, 640), device=dev).half()
m = bench.build_model("yolov5s", dev)
with torch.no_grad():
    for _ in range(n):
        m(x)
torch.cuda.synchronize()
print("forwards", n)
