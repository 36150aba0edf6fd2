"""Round 5 diagnostic: Y5_DETERMINISTIC=1 training step (yolov5s, 64 x 3 x 640 x 640 fp16, 512 targets) run several times -- which parameter gradients differ between
runs, by how much, and which kernels computed them (forward cfg, data-gradient cfgs, weight-gradient choice of the layer)."""
import os
import sys

import torch

sys.path.insert(0, ".")
os.environ["Y5_DETERMINISTIC"] = "1"
from oracle import detgen  # noqa: E402
from tests.test_gpu_train import _model, _step_grads  # noqa: E402
from yolov5_amd.loss import ComputeLoss  # noqa: E402

dev = torch.device("cuda:0")
B, S, SCALE = 64, 640, 1024.0
g = torch.Generator().manual_seed(11)
x = torch.rand((B, 3, S, S), generator=g).half().to(dev)
t = torch.from_numpy(detgen.synth_targets(B, 8, seed=11)).to(dev)
m, cfg, sd = _model("yolov5s", dev)
cl = ComputeLoss(m)
runs = [_step_grads(m, cl, x, t, SCALE) for _ in range(int(os.environ.get("RUNS", "4")))]
eng = next(iter(m.__dict__["_train_engines"].values()))
plan = {st["op"]["name"]: (st["fcfg"], [c for sub in st["subs"] for c in sub["cfg"].values()], st.get("wg_choice")) for st in eng.convs}
names = [n for n, _ in m.named_parameters()]
print("losses", [r[0] for r in runs])
for k in range(1, len(runs)):
    diff = []
    for n in names:
        a, b = runs[0][2][n], runs[k][2][n]
        if not torch.equal(a, b):
            d = (a - b).abs()
            diff.append((n, int((a != b).sum()), a.numel(), float(d.max()), float(a.abs().max())))
    print(f"run {k} vs 0: {len(diff)} of {len(names)} parameters differ")
    for row in diff[:60]:
        print("   ", row)
print("plan:")
for k, v in plan.items():
    print("   ", k, v)
