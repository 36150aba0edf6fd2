#!/usr/bin/env python
"""Exactly N forwards of the headline configuration (yolov5s, 64 x 3x640x640 fp16) IN THE MODE bench.py TIMES (Detect.export = True: the fused
conv + decode heads, no raw-logits output), nothing else on the GPU -- the target of the PMC passes of scripts/pmc_forward.sh / pmc_issue_mix.sh
(run once with Y5_TUNE_CACHE set to fill the tile-choice cache first).  Writes the plan it ran ([op, configuration] in launch order, and the launches per
kernel family) to gpurun_out/forward_only_plan.json: the PMC scripts stamp it into their results, and bench.py reports counter figures only for a timed
plan with the same kernel families x launches (round 5's counters had profiled the non-export plan)."""
import json
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
m.model[-1].export = True   # as bench.py
with torch.no_grad():
    for _ in range(n):
        m(x)
torch.cuda.synchronize()
eng_top = next(iter(m._engines.values()))
eng = eng_top.engines[0] if getattr(eng_top, "parts", 1) > 1 else eng_top
plan = [[nm, c] for nm, c in eng.plan_table()]
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/forward_only_plan.json", "w") as f:
    json.dump({"plan": plan, "plan_sha16": bench.plan_hash(plan), "plan_families": bench.plan_families(plan)}, f)
print("forwards", n)
