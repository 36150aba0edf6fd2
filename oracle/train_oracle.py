"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (yolov5_amd/).

CPU restatement of the reference's training loop over the oracle model (oracle/yolo_oracle.py), torch-CPU fp32 autograd:
train.py:234-248 (nominal batch 64 -> accumulate / weight-decay scaling, three-group SGD-Nesterov, LambdaLR), :340-434 (warm-up
interpolation, forward in train mode, ComputeLoss, loss *= WORLD_SIZE, backward, clip_grad_norm_(10), optimizer step, EMA update,
scheduler.step) and utils/torch_utils.py:257-290 (`smart_optimizer` groups), :343-365 (`ModelEMA`).  No GradScaler (fp32).
Pinned against a loop built from the reference's OWN pieces in tests/test_oracle_vs_reference.py."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import yolo_oracle as yo

HYP = {"lr0": 0.01, "lrf": 0.01, "momentum": 0.937, "weight_decay": 0.0005, "warmup_epochs": 3.0, "warmup_momentum": 0.8,
       "warmup_bias_lr": 0.1, **yo.HYP_SCRATCH_LOW}  # data/hyps/hyp.scratch-low.yaml


def param_groups(sd):
    """utils/torch_utils.py:257-276 in state_dict terms: g2 = biases (module parameter named 'bias'), g1 = BatchNorm weights,
    g0 = everything else (decayed).  Returns lists of keys in module-registration order."""
    g0, g1, g2 = [], [], []
    for k, v in sd.items():
        if not v.dtype.is_floating_point or k.endswith(("running_mean", "running_var", "anchors")):
            continue
        if k.endswith(".bias"):
            g2.append(k)
        elif ".bn." in k and k.endswith(".weight"):
            g1.append(k)
        else:
            g0.append(k)
    return g0, g1, g2


def train_oracle(cfg, sd, imgs, targets_per_image, batch_size, hyp=None, epochs=1, cos_lr=False, nbs=64, world=1, rank=-1):
    """imgs: (N, 3, H, W) uint8 or float; targets_per_image: list of (k, 6) tensors (column 0 ignored).  Returns dict(losses (steps, 3),
    lr (epochs, 3 groups in optimizer order g2, g0, g1), sd (trained leaves), ema (state dict), updates)."""
    hyp = dict(HYP if hyp is None else hyp)
    sd = {k: v.clone() for k, v in sd.items()}
    g0, g1, g2 = param_groups(sd)
    for k in g0 + g1 + g2:
        sd[k].requires_grad_(True)
    idx = list(range(imgs.shape[0]))
    if rank != -1:
        idx = idx[rank::world]
    nb = (len(idx) + batch_size - 1) // batch_size
    total_batch = batch_size * world if rank != -1 else batch_size
    accumulate = max(round(nbs / total_batch), 1)
    hyp["weight_decay"] *= total_batch * accumulate / nbs
    opt = torch.optim.SGD([sd[k] for k in g2], lr=hyp["lr0"], momentum=hyp["momentum"], nesterov=True)       # torch_utils.py:283
    opt.add_param_group({"params": [sd[k] for k in g0], "weight_decay": hyp["weight_decay"]})                 # :287
    opt.add_param_group({"params": [sd[k] for k in g1], "weight_decay": 0.0})                                 # :288
    lf = (lambda x: ((1 - math.cos(x * math.pi / epochs)) / 2) * (hyp["lrf"] - 1) + 1) if cos_lr else \
        (lambda x: (1 - x / epochs) * (1.0 - hyp["lrf"]) + hyp["lrf"])
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lf)
    ema = {k: v.detach().clone() for k, v in sd.items()}
    updates = 0
    decay = lambda u: 0.9999 * (1 - math.exp(-u / 2000))  # noqa: E731  torch_utils.py:350
    anchors = yo.model_anchors(cfg)
    nw = max(round(hyp["warmup_epochs"] * nb), 100)
    last_opt_step = -1
    losses, lrs = [], []
    for epoch in range(epochs):
        opt.zero_grad()
        for i in range(nb):
            ids = idx[i * batch_size:(i + 1) * batch_size]
            ni = i + nb * epoch
            x = imgs[ids]
            x = x.float() / 255 if x.dtype == torch.uint8 else x.float()
            t = []
            for k, j in enumerate(ids):
                tj = targets_per_image[j].clone().float()
                tj[:, 0] = k
                t.append(tj)
            t = torch.cat(t, 0)
            if ni <= nw:
                xi = [0, nw]
                accumulate = max(1, np.interp(ni, xi, [1, nbs / total_batch]).round())
                for j, g in enumerate(opt.param_groups):
                    g["lr"] = float(np.interp(ni, xi, [hyp["warmup_bias_lr"] if j == 0 else 0.0, g["initial_lr"] * lf(epoch)]))
                    g["momentum"] = float(np.interp(ni, xi, [hyp["warmup_momentum"], hyp["momentum"]]))
            pred = yo.model_forward(cfg, sd, x, training=True, bn_batch_stats=True)
            loss, items = yo.compute_loss(pred, t, anchors, hyp)
            if rank != -1:
                loss = loss * world
            loss.backward()
            if ni - last_opt_step >= accumulate:
                torch.nn.utils.clip_grad_norm_([sd[k] for k in g0 + g1 + g2], max_norm=10.0)
                opt.step()
                opt.zero_grad()
                updates += 1
                d = decay(updates)
                with torch.no_grad():
                    for k, v in ema.items():
                        if v.dtype.is_floating_point:
                            v.mul_(d).add_((1 - d) * sd[k].detach())
                last_opt_step = ni
            losses.append(items.detach().clone())
        lrs.append([g["lr"] for g in opt.param_groups])
        sched.step()
    return dict(losses=torch.stack(losses), lr=lrs, sd={k: v.detach() for k, v in sd.items()}, ema=ema, updates=updates)


def synthetic_set(n, hw, per_img=3, seed=0):
    """A fixed synthetic training set: structured uint8 images (detgen.scene) and `per_img` boxes per image."""
    from . import detgen

    imgs = torch.from_numpy((detgen.scene((n, 3, hw, hw), seed=seed) * 255).round().astype(np.uint8))
    t = torch.from_numpy(detgen.synth_targets(n, per_img, seed=seed))
    return imgs, [t[t[:, 0] == i].clone() for i in range(n)]
