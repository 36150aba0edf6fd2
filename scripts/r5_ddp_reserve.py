#!/usr/bin/env python
"""Round-5 aid (GPU): exposed cost of the gradient exchange through a ONE-rank RCCL group as a function of the CUs the backward plan leaves to the
collective (HipDDP.reserve_cus -> y5_set_cu_budget, csrc/core.hip).  Prints the plain step and, per reservation, the step through smart_DDP."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
plain = bench.train_probe("yolov5s", 64, 640, dev, 1, steps=20, warmup=5)
print("plain", plain["ms_per_step"], plain["step_ms"]["median"], flush=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for r in (sys.argv[1:] or ["0", "8", "16", "32", "64", "0", "16"]):
    if r.startswith("dry"):      # bookkeeping without the collective calls
        os.environ["Y5_EXPERIMENTAL"], r = "ddp_dry", r[3:] or "0"
    else:
        os.environ["Y5_EXPERIMENTAL"] = ""
    os.environ["Y5_DDP_SYNC"] = "last"
    for m in ("all", "none", "last"):
        if r.startswith(m):
            os.environ["Y5_DDP_SYNC"], r = m, r[len(m):] or "0"
    if r.startswith("cap"):      # one bucket
        os.environ["Y5_DDP_BUCKET_MB"], r = "1000", r[3:] or "0"
    else:
        os.environ["Y5_DDP_BUCKET_MB"] = "6"
    os.environ["Y5_DDP_RESERVE_CUS"] = r
    ex = bench.train_probe("yolov5s", 64, 640, dev, 1, steps=20, warmup=5, exchange_group=True)
    print(f"dry={os.environ['Y5_EXPERIMENTAL']} sync={os.environ['Y5_DDP_SYNC']} reserve {r} CUs: buckets {ex['allreduce_buckets']}  step {ex['step_ms']['median']:.3f} ms  exposed {1e3 * (ex['step_ms']['median'] - plain['step_ms']['median']):.0f} us", flush=True)
dist.destroy_process_group()
