#!/usr/bin/env python
"""Round-5 aid (GPU): fp16 training plan against the fp32 plan of the same step (yolov5s bs 64 640^2): median / worst relative L2 distance of the 177 parameter
gradients -- a regression in one backward kernel shows as a jump from ~0.1 (fp16 storage noise) to ~0.5.  Environment switches select what runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import detgen, yolo_oracle as yo
from yolov5_amd.loss import ComputeLoss
from yolov5_amd.yolo import DetectionModel

dev = torch.device("cuda:0")
B, S = int(os.environ.get("BISECT_B", "64")), 640
cfg = yo.model_cfg("yolov5s")
sd = yo.det_state_dict(cfg, 0, fused=False)
m = DetectionModel("yolov5s.yaml")
m.load_state_dict(sd)
m.hyp = dict(yo.HYP_SCRATCH_LOW)
m = m.to(dev).train()
cl = ComputeLoss(m)
g = torch.Generator().manual_seed(11)
x = torch.rand((B, 3, S, S), generator=g)
t = torch.from_numpy(detgen.synth_targets(B, 8, seed=11)).to(dev)


def grads(xx, scale):
    for p in m.parameters():
        p.grad = None
    loss, _ = cl(m(xx), t)
    (loss * scale).backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.float().cpu().flatten().double() / scale for n, p in m.named_parameters()}


l32, g32 = grads(x.to(dev), 1.0)
m.__dict__["_train_engines"].clear()
l16, g16 = grads(x.half().to(dev), float(os.environ.get("BISECT_SCALE", "1024")))
eng = next(iter(m.__dict__["_train_engines"].values()))
plan = []
for st in eng.convs:
    plan.append((st["op"]["name"], st["fcfg"], [tuple(sorted(sub["cfg"].items())) for sub in st["subs"]], st.get("wg_choice")))
if os.environ.get("BISECT_PLAN"):
    import json
    json.dump(plan, open(os.environ["BISECT_PLAN"], "w"))
rel = {n: float((g16[n] - g32[n]).norm() / (g32[n].norm() + 1e-30)) for n in g32}
v = np.array(list(rel.values()))
worst = sorted(rel.items(), key=lambda kv: -kv[1])[:5]
print(f"{os.environ.get('BISECT_TAG', '')}: loss fp32 {l32:.5f} fp16 {l16:.5f}; gradient rel-L2 fp16 vs fp32 plan: median {np.median(v):.4f} p90 {np.percentile(v, 90):.4f} worst {v.max():.4f}  {[(n, round(r, 3)) for n, r in worst]}")

if os.environ.get("BISECT_CHAOS"):
    # sensitivity of the step to a perturbation far below any kernel's rounding: ONE input value moved by one fp16 ulp
    xp = x.half().clone()
    xp[0, 0, 0, 0] = torch.nextafter(xp[0, 0, 0, 0].float(), torch.tensor(2.0)).half() if False else (xp[0, 0, 0, 0].float() * (1 + 2.0 ** -10)).half()
    _, g16b = grads(xp.to(dev), 1024.0)
    r2 = np.array([float((g16b[n] - g16[n]).norm() / (g16[n].norm() + 1e-30)) for n in g16])
    print(f"chaos fp16 plan: one input value + 1 ulp -> gradient rel-L2 change median {np.median(r2):.4f} p90 {np.percentile(r2, 90):.4f} worst {r2.max():.4f}")
    m.__dict__["_train_engines"].clear()
    _, g32b = grads(xp.float().to(dev), 1.0)
    _, g32a = grads(x.half().float().to(dev), 1.0)
    r3 = np.array([float((g32b[n] - g32a[n]).norm() / (g32a[n].norm() + 1e-30)) for n in g32a])
    print(f"chaos fp32 plan: one input value + 1 ulp -> gradient rel-L2 change median {np.median(r3):.4f} p90 {np.percentile(r3, 90):.4f} worst {r3.max():.4f}")
