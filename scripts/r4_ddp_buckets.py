#!/usr/bin/env python
"""Round-4 aid (GPU): exposed cost of the gradient exchange through a ONE-rank RCCL group as a function of the bucket cap (HipDDP buckets of the flat
gradient arena).  Prints the plain step and, per cap, the step through smart_DDP."""
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
for cap in (sys.argv[1:] or ["6", "12", "16", "32", "6"]):
    os.environ["Y5_DDP_BUCKET_MB"] = cap
    ex = bench.train_probe("yolov5s", 64, 640, dev, 1, steps=20, warmup=5, exchange_group=True)
    print(f"cap {cap} MB: buckets {ex['allreduce_buckets']}  step {ex['step_ms']['median']:.3f} ms  exposed {1e3 * (ex['step_ms']['median'] - plain['step_ms']['median']):.0f} us", flush=True)
dist.destroy_process_group()
