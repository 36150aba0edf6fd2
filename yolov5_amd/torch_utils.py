"""Training-side helpers with the reference's names (utils/torch_utils.py): `smart_DDP` :61-70, `de_parallel`,
`torch_distributed_zero_first` :96-105, `smart_optimizer` :257-290, `ModelEMA` :343-375.

`smart_DDP` does NOT wrap the model in torch's DistributedDataParallel: the whole network backward is one autograd node
here (yolov5_amd/train_engine.py), so DDP's per-parameter autograd hooks could only fire after the last kernel and nothing
would overlap.  `HipDDP` is driven by the backward PLAN instead: the training engine reports every parameter gradient the
moment its kernels are queued (reverse layer order), gradients are packed into flat fp32 buckets and each full bucket is
all-reduced (RCCL over xGMI; `nccl` backend) asynchronously on the process group's stream while the remaining backward
kernels run on the compute stream.  Semantics = the reference's: gradients are AVERAGED over ranks and train.py:404-405
multiplies the loss by WORLD_SIZE, i.e. the applied gradient is the sum over ranks of the per-rank batch gradients;
parameters and buffers are broadcast from rank 0 at construction; BatchNorm statistics stay per rank (no SyncBN).
"""
from __future__ import annotations

import math
from contextlib import contextmanager
from copy import deepcopy

import torch
import torch.distributed as dist
from torch import nn


def is_parallel(model):
    return type(model).__name__ in ("DataParallel", "DistributedDataParallel", "HipDDP")


def de_parallel(model):
    return model.module if is_parallel(model) else model


class _Bucket:
    def __init__(self, idxs, numels, device):
        self.idxs = idxs
        self.offsets = {}
        off = 0
        for i, n in zip(idxs, numels):
            self.offsets[i] = (off, n)
            off += n
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.pending = len(idxs)
        self.work = None


class HipDDP(nn.Module):
    """Data-parallel wrapper for yolov5_amd models (see module docstring).  bucket_cap_mb follows torch DDP's default
    (25 MB: yolov5s = 28.9 MB of fp32 gradients -> 2 buckets; a ring all-reduce of one bucket is ~0.3 ms on one xGMI link)."""

    def __init__(self, module, bucket_cap_mb=25.0, process_group=None):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = list(module.parameters())
        with torch.no_grad():
            if self.world > 1:
                for t in list(module.parameters()) + [b for b in module.buffers() if b.dtype.is_floating_point or b.dtype == torch.long]:
                    dist.broadcast(t.data, 0, group=process_group)
        cap = int(bucket_cap_mb * 1024 * 1024 / 4)
        self.buckets, cur, cur_n = [], [], 0
        dev = self.params[0].device
        for i in reversed(range(len(self.params))):  # gradients arrive in (roughly) reverse registration order
            n = self.params[i].numel()
            if cur and cur_n + n > cap:
                self.buckets.append(_Bucket(cur, [self.params[j].numel() for j in cur], dev))
                cur, cur_n = [], 0
            cur.append(i)
            cur_n += n
        if cur:
            self.buckets.append(_Bucket(cur, [self.params[j].numel() for j in cur], dev))
        self.p2b = {i: b for b in self.buckets for i in b.idxs}
        module.__dict__["_ddp_sink"] = self
        for k in ("stride", "names", "hyp", "nc", "yaml"):
            if hasattr(module, k):
                setattr(self, k, getattr(module, k))

    def forward(self, *a, **k):
        return self.module(*a, **k)

    # ---- gradient sink protocol (called by TrainEngine.backward) -------------------------------------------------------
    def begin(self):
        for b in self.buckets:
            b.pending = len(b.idxs)
            b.work = None

    def _launch(self, b):
        if self.world > 1:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            b.work = None
        b.pending = -1

    def grad_ready(self, idx, g):
        """Gradient of parameter `idx` is queued on the compute stream: pack it; a full bucket goes on the wire at once."""
        b = self.p2b[idx]
        off, n = b.offsets[idx]
        view = b.flat[off:off + n].view(self.params[idx].shape)
        view.copy_(g)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)
        return view

    def finish(self, grads):
        """All kernels of the backward plan are queued: launch stragglers, wait for the wire, average."""
        for b in self.buckets:
            if b.pending > 0:  # parameters that received no gradient this step contribute zeros
                for i in b.idxs:
                    if grads[i] is None:
                        off, n = b.offsets[i]
                        b.flat[off:off + n].zero_()
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
            if self.world > 1:
                b.flat.div_(self.world)
        return grads


def smart_DDP(model):
    """utils/torch_utils.py:61-70 (one process per GPU under torchrun; `nccl` = RCCL on ROCm)."""
    return HipDDP(model)


@contextmanager
def torch_distributed_zero_first(local_rank: int):
    """utils/torch_utils.py:96-105."""
    if local_rank not in [-1, 0]:
        dist.barrier()
    yield
    if local_rank == 0:
        dist.barrier()


def smart_optimizer(model, name="SGD", lr=0.01, momentum=0.937, decay=5e-4):
    """utils/torch_utils.py:257-290: 3 parameter groups -- biases (no decay), BatchNorm weights (no decay), other weights (decay)."""
    g = [], [], []
    bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for v in model.modules():
        for p_name, p in v.named_parameters(recurse=0):
            if p_name == "bias":
                g[2].append(p)
            elif p_name == "weight" and isinstance(v, bn):
                g[1].append(p)
            else:
                g[0].append(p)
    if name == "Adam":
        optimizer = torch.optim.Adam(g[2], lr=lr, betas=(momentum, 0.999))
    elif name == "AdamW":
        optimizer = torch.optim.AdamW(g[2], lr=lr, betas=(momentum, 0.999), weight_decay=0.0)
    elif name == "RMSProp":
        optimizer = torch.optim.RMSprop(g[2], lr=lr, momentum=momentum)
    elif name == "SGD":
        optimizer = torch.optim.SGD(g[2], lr=lr, momentum=momentum, nesterov=True)
    else:
        raise NotImplementedError(f"Optimizer {name} not implemented.")
    optimizer.add_param_group({"params": g[0], "weight_decay": decay})
    optimizer.add_param_group({"params": g[1], "weight_decay": 0.0})
    return optimizer


class ModelEMA:
    """utils/torch_utils.py:343-375: EMA of parameters AND buffers, fp32, decay ramp 1 - exp(-updates / tau)."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = deepcopy(de_parallel(model)).eval()
        self.ema.__dict__.pop("_train_engines", None)
        self.ema.__dict__.pop("_ddp_sink", None)
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        self.updates += 1
        d = self.decay(self.updates)
        msd = de_parallel(model).state_dict()
        for k, v in self.ema.state_dict().items():
            if v.dtype.is_floating_point:
                v *= d
                v += (1 - d) * msd[k].detach()
