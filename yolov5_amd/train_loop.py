"""The reference's training loop, composed from the yolov5_amd seams (train.py:225-258 set-up, :340-434 epoch / batch loop).
Not the CLI: no argparse, datasets, loggers or callbacks -- the schedule arithmetic and the step order, so that a model TRAINS the
way train.py trains it:

  nominal batch 64 -> `accumulate`, weight-decay scaling (:234-236); three-group SGD-Nesterov (`smart_optimizer`);
  linear / cosine epoch schedule as a LambdaLR (:239-248); EMA on rank 0 (:251); per batch (:372-421):
  warm-up interpolation of lr / momentum / accumulate over the first max(warmup_epochs * nb, 100) iterations,
  uint8 -> float / 255 (inside the engine), forward, ComputeLoss, loss *= WORLD_SIZE under DDP, scaled backward,
  every `accumulate` batches: unscale + clip_grad_norm_(10) + optimizer step + scaler update + zero_grad + EMA update;
  per epoch: scheduler.step().

MI355X mapping: forward / backward are the TrainEngine plans, the optimizer block is three fused launches (HipSGD.step_fused), the
GradScaler's found-inf flag stays on the device -- the update it guards is skipped there (csrc/optim.hip) and the host reads the
flag with a non-blocking copy one iteration later to adjust the scale (no pipeline drain per step).
"""
from __future__ import annotations

import math

from copy import deepcopy

import numpy as np
import torch
import torch.distributed as dist

from .loss import ComputeLoss
from .torch_utils import ModelEMA, de_parallel, smart_DDP, smart_optimizer

HYP_SCRATCH_LOW = {  # data/hyps/hyp.scratch-low.yaml
    "lr0": 0.01, "lrf": 0.01, "momentum": 0.937, "weight_decay": 0.0005, "warmup_epochs": 3.0, "warmup_momentum": 0.8,
    "warmup_bias_lr": 0.1, "box": 0.05, "cls": 0.5, "cls_pw": 1.0, "obj": 1.0, "obj_pw": 1.0, "iou_t": 0.2, "anchor_t": 4.0,
    "fl_gamma": 0.0, "label_smoothing": 0.0,
}


def one_cycle(y1=0.0, y2=1.0, steps=100):
    """Cosine ramp from y1 to y2 over `steps` (the reference imports it from ultralytics; used at train.py:240)."""
    return lambda x: ((1 - math.cos(x * math.pi / steps)) / 2) * (y2 - y1) + y1


def lr_lambda(epochs, lrf, cos_lr=False):
    """train.py:239-246."""
    if cos_lr:
        return one_cycle(1, lrf, epochs)
    return lambda x: (1 - x / epochs) * (1.0 - lrf) + lrf


class LossScaler:
    """torch.cuda.amp.GradScaler's policy (init 65536, x2 after 2000 clean steps, x0.5 on overflow) with the overflow flag read
    asynchronously: `record(stats)` queues a non-blocking copy of the fused step's [norm, coef, found_inf, -] vector, the next
    `update()` that finds it complete applies the policy.  enabled=False: scale 1, nothing to do (fp32 training)."""

    def __init__(self, enabled=True, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.enabled = enabled
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._clean = 0
        self._pending = []
        self.skipped = 0

    def record(self, stats):
        if not self.enabled or stats is None:
            return
        if stats.is_cuda:
            host = torch.empty(4, dtype=torch.float32, pin_memory=True)
            host.copy_(stats, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((host, ev))
        else:
            self._pending.append((stats.clone(), None))

    def update(self, block=False):
        while self._pending and (block or self._pending[0][1] is None or self._pending[0][1].query()):
            host, ev = self._pending.pop(0)
            if ev is not None:
                ev.synchronize()
            if float(host[2]) != 0.0:
                self.scale *= self.backoff_factor
                self._clean = 0
                self.skipped += 1
            else:
                self._clean += 1
                if self._clean >= self.growth_interval:
                    self.scale *= self.growth_factor
                    self._clean = 0


def host_amp_step(optimizer, parameters, scaler, max_norm=10.0):
    """GradScaler.unscale_ + clip_grad_norm_ + GradScaler.step for ANY torch optimizer (the Adam / AdamW / RMSProp choices of
    smart_optimizer; train.py:411-414): unscale in place, clip, and SKIP optimizer.step() when the unscaled gradients are not finite;
    the flag goes to the scale policy (back-off / growth at the next update()).  Returns True when the step was taken.
    One host sync (the norm) -- this is the non-fused path; HipSGD.step_fused keeps the flag on the device."""
    params = [p for p in parameters if p.grad is not None]
    if scaler.scale != 1.0 and params:
        torch._foreach_mul_([p.grad for p in params], 1.0 / scaler.scale)
    total = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm) if params else torch.zeros(())
    found_inf = not bool(torch.isfinite(total))
    if not found_inf:
        optimizer.step()
    scaler.record(torch.tensor([float("inf") if found_inf else float(total), 0.0, float(found_inf), 0.0]))
    return not found_inf


def train(model, loader, hyp=None, epochs=1, device=None, batch_size=None, cos_lr=False, amp=True, ema=True, world_size=1, rank=-1,
          optimizer_name="SGD", max_norm=10.0, start_epoch=0, on_batch_end=None, nbs=64, val_loader=None, noval=False, sync_bn=False):
    """Runs `epochs` epochs over `loader` (iterable of (imgs uint8|float BCHW, targets (nt, 6), *rest), re-iterable, len() = batches
    per epoch).  Returns dict(model, ema, optimizer, scheduler, scaler, mloss per epoch, losses per iteration, lr per epoch).
    rank / world_size as train.py's RANK / WORLD_SIZE (-1 / 1: single process; otherwise torch.distributed is initialised and the
    model is wrapped by smart_DDP).  val_loader: validated once per epoch on rank -1 / 0 with the EMA model (train.py:440-455 -> val_loop.run;
    noval: only after the final epoch); sync_bn: train.py:269-271 (`--sync-bn` under DDP: BatchNorm statistics over the global batch); `results` per validated epoch = (P, R, mAP@.5, mAP@.5:.95, val box / obj / cls loss), `fitness` beside it."""
    hyp = dict(HYP_SCRATCH_LOW if hyp is None else hyp)
    device = torch.device(device) if device is not None else next(model.parameters()).device
    nb = len(loader)
    if batch_size is None:
        batch_size = int(next(iter(loader))[0].shape[0])
    total_batch = batch_size * world_size if rank != -1 else batch_size  # train.py works with the TOTAL batch size here
    accumulate = max(round(nbs / total_batch), 1)                                  # :235
    hyp["weight_decay"] *= total_batch * accumulate / nbs                           # :236
    optimizer = smart_optimizer(model, optimizer_name, hyp["lr0"], hyp["momentum"], hyp["weight_decay"])  # :237
    lf = lr_lambda(epochs, hyp["lrf"], cos_lr)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lf)          # :248
    ema_obj = ModelEMA(model) if (ema and rank in (-1, 0)) else None                # :251
    if sync_bn and rank != -1:                                                      # :269-271
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).to(device)
    model.hyp = hyp                                                                 # :331
    ddp = smart_DDP(model) if rank != -1 and world_size > 1 else model              # :322
    compute_loss = ComputeLoss(model)                                               # :352
    nw = max(round(hyp["warmup_epochs"] * nb), 100)                                 # :341
    last_opt_step = -1
    scheduler.last_epoch = start_epoch - 1                                          # :347
    scaler = LossScaler(enabled=amp)
    fused = hasattr(optimizer, "step_fused")
    hist = {"losses": [], "mloss": [], "lr": [], "results": [], "fitness": []}
    best_fitness = 0.0
    for epoch in range(start_epoch, epochs):
        model.train()
        mloss = torch.zeros(3, device=device)
        if rank != -1 and hasattr(getattr(loader, "sampler", None), "set_epoch"):
            loader.sampler.set_epoch(epoch)                                         # :365
        optimizer.zero_grad()
        for i, batch in enumerate(loader):
            imgs, targets = batch[0], batch[1]
            ni = i + nb * epoch                                                     # :374
            imgs = imgs.to(device, non_blocking=True)  # uint8 stays uint8: the engine's input kernel does the float() / 255 of :375
            if ni <= nw:                                                            # warm-up :378-387
                xi = [0, nw]
                accumulate = max(1, np.interp(ni, xi, [1, nbs / total_batch]).round())
                for j, g in enumerate(optimizer.param_groups):
                    g["lr"] = float(np.interp(ni, xi, [hyp["warmup_bias_lr"] if j == 0 else 0.0, g["initial_lr"] * lf(epoch)]))
                    if "momentum" in g:
                        g["momentum"] = float(np.interp(ni, xi, [hyp["warmup_momentum"], hyp["momentum"]]))
            if amp:   # fp16 plan: uint8 goes in as it is (the input kernel divides by 255), floats as fp16
                x = imgs if imgs.dtype == torch.uint8 else imgs.half()
            else:     # fp32 plan (train.py without AMP): float32 images select it (train_engine.train_forward)
                x = imgs.float() / 255 if imgs.dtype == torch.uint8 else imgs.float()
            pred = ddp(x)                                                           # :399
            loss, loss_items = compute_loss(pred, targets.to(device))               # :400
            if rank != -1:
                loss = loss * world_size                                            # :401-402
            # the scale is FROZEN for all micro-batches of an accumulation window (GradScaler only changes it in update(), right after
            # an optimizer step, train.py:414): pending growth / back-off is applied below, behind zero_grad, never between two backwards
            (loss * scaler.scale).backward()                                        # :407
            if ni - last_opt_step >= accumulate:                                    # :410-419
                if fused:
                    stats = optimizer.step_fused(inv_scale=1.0 / scaler.scale, max_norm=max_norm, ema=ema_obj, model=model)
                    scaler.record(stats)
                else:
                    host_amp_step(optimizer, model.parameters(), scaler, max_norm)
                    if ema_obj is not None:
                        ema_obj.update(model)
                optimizer.zero_grad()
                scaler.update()   # window boundary: the only place the scale may change
                last_opt_step = ni
            mloss = (mloss * i + loss_items) / (i + 1)                              # :423
            hist["losses"].append(loss_items.detach())
            if on_batch_end is not None:
                on_batch_end(ni, loss_items, optimizer)
        hist["lr"].append([g["lr"] for g in optimizer.param_groups])                # :432
        scheduler.step()                                                            # :433
        hist["mloss"].append(mloss.detach())
        if val_loader is not None and rank in (-1, 0):                              # :440-455
            final_epoch = epoch + 1 == epochs
            if not noval or final_epoch:
                from . import val_loop
                from .metrics import fitness

                # train.py:443 validates `ema.ema`; val.py:187,388 converts that copy to fp16 and back IN PLACE.  Without an EMA the live
                # model must not take that round trip (fp16 rounding of the fp32 master weights every epoch): validate a copy of it.
                vm = ema_obj.ema if ema_obj is not None else deepcopy(de_parallel(ddp))
                results, _maps, _ = val_loop.run(vm, val_loader, half=amp, compute_loss=compute_loss)
                if ema_obj is not None:
                    ema_obj.invalidate()   # `.half()` / `.float()` re-created the EMA's buffer tensors (ModelEMA also checks by itself)
                fi = float(fitness(np.array(results).reshape(1, -1))[0])            # :457 weighted [P, R, mAP@.5, mAP@.5:.95]
                best_fitness = max(best_fitness, fi)
                hist["results"].append(tuple(float(v) for v in results))
                hist["fitness"].append(fi)
    scaler.update(block=True)
    hist["losses"] = torch.stack(hist["losses"]).float().cpu() if hist["losses"] else torch.zeros(0, 3)
    hist["mloss"] = torch.stack(hist["mloss"]).float().cpu() if hist["mloss"] else torch.zeros(0, 3)
    return dict(model=de_parallel(ddp), ema=ema_obj, optimizer=optimizer, scheduler=scheduler, scaler=scaler, best_fitness=best_fitness, **hist)


def pad_to_common(idx, n, world_size):
    """DistributedSampler's non-drop_last rule (torch.utils.data.distributed; utils/dataloaders.py:94-101): every rank's index list is
    extended to num_samples = ceil(n / world) by repeating its own head, so all ranks see the same number of batches -- a rank with
    one batch more would wait forever in its gradient all-reduce."""
    num = (n + world_size - 1) // world_size
    idx = list(idx)
    pad = num - len(idx)
    if pad > 0 and idx:
        idx += (idx * math.ceil(pad / len(idx)))[:pad]
    return idx


class TensorLoader:
    """Minimal re-iterable loader over an in-memory set: batches of (imgs, targets, paths, shapes) like the reference's collate_fn
    output (utils/dataloaders.py:858-863: column 0 of targets = image index inside the batch)."""

    def __init__(self, imgs, targets_per_image, batch_size, rank=-1, world_size=1):
        self.imgs, self.tpi, self.bs = imgs, targets_per_image, batch_size
        n = imgs.shape[0]
        idx = list(range(n))
        if rank != -1:  # SmartDistributedSampler: a fixed disjoint subset per rank (dataloaders.py:79-103) ...
            idx = pad_to_common(idx[rank::world_size], n, world_size)  # ... padded so that every rank runs the same number of batches
        self.idx = idx

    def __len__(self):
        return (len(self.idx) + self.bs - 1) // self.bs

    def __iter__(self):
        for b in range(len(self)):
            ids = self.idx[b * self.bs:(b + 1) * self.bs]
            t = []
            for k, i in enumerate(ids):
                ti = self.tpi[i].clone()
                ti[:, 0] = k
                t.append(ti)
            yield self.imgs[ids], torch.cat(t, 0) if t else torch.zeros(0, 6), [f"img{i}" for i in ids], None
