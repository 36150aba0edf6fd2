#!/usr/bin/env python
"""Training-step benchmark (BASELINE config 3, per-GPU shape): yolov5s, 64 x 3x640x640 fp16, 512 synthetic targets;
one step = forward (train-mode BN) + ComputeLoss + backward (+ RCCL gradient all-reduce under torchrun) + SGD.
    python scripts/train_bench.py --steps 10                       # one GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/train_bench.py   # N GPUs"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolov5s")
    a = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl")
    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.torch_utils import ModelEMA, smart_DDP, smart_optimizer
    from yolov5_amd.yolo import DetectionModel

    torch.manual_seed(0)
    m = DetectionModel(a.model + ".yaml").to(dev).train()
    m.hyp = {"box": 0.05, "cls": 0.5, "cls_pw": 1.0, "obj": 1.0, "obj_pw": 1.0, "anchor_t": 4.0, "fl_gamma": 0.0, "label_smoothing": 0.0}
    compute_loss = ComputeLoss(m)
    model = smart_DDP(m) if world > 1 else m
    opt = smart_optimizer(m, "SGD", lr=0.01, momentum=0.937, decay=5e-4)  # HipSGD: 3 groups, fused multi-tensor step
    ema = ModelEMA(m)
    g = torch.Generator(device="cpu").manual_seed(rank)
    x = torch.rand((a.batch, 3, a.imgsz, a.imgsz), generator=g).half().to(dev)
    nt = a.batch * 8
    t = torch.cat((torch.randint(0, a.batch, (nt, 1), generator=g).float(), torch.randint(0, 80, (nt, 1), generator=g).float(),
                   torch.rand((nt, 2), generator=g) * 0.8 + 0.1, torch.rand((nt, 2), generator=g) * 0.3 + 0.02), 1).to(dev)
    scale = 1024.0

    def step():
        pred = model(x)
        loss, _ = compute_loss(pred, t)
        if world > 1:
            loss = loss * world  # train.py:404-405
        opt.zero_grad(set_to_none=True)
        (loss * scale).backward()
        # train.py:413-421 scaler.unscale_ + clip_grad_norm_(10.0) + optimizer step + ema.update, fused (csrc/optim.hip)
        opt.step_fused(inv_scale=1.0 / scale, max_norm=10.0, ema=ema, model=m)
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "training images/sec (yolov5s, 64 img/GPU, 640px, fp16 AMP-style)", "value": round(a.batch * world * a.steps / dt, 1),
                          "unit": "images/sec", "n_gpus": world, "ms_per_step": round(dt / a.steps * 1e3, 2), "loss": round(float(loss.detach()), 4)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
