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
import os
from contextlib import contextmanager
from copy import deepcopy

import torch
import torch.distributed as dist
from torch import nn

from . import _lib as _libm, _state


def is_parallel(model):
    return type(model).__name__ in ("DataParallel", "DistributedDataParallel", "HipDDP")


def de_parallel(model):
    return model.module if is_parallel(model) else model


class _Bucket:
    """A contiguous range [lo, hi) of the training engine's gradient arena and the parameters that live in it."""

    def __init__(self, idxs, lo, hi):
        self.idxs, self.lo, self.hi = idxs, lo, hi
        self.pending = len(idxs)
        self.work = None


class HipDDP(nn.Module):
    """Data-parallel wrapper for yolov5_amd models (see module docstring).  Buckets are ranges of the engine's flat gradient
    arena (train_engine.py): nothing is packed or copied.  bucket_cap_mb = 6: yolov5s' 28.9 MB of fp32 gradients make 5 buckets,
    so the first all-reduce is on the wire after the head + last C3 of the backward plan instead of after half of it (torch DDP's
    25 MB default would give 2); a 6 MB ring all-reduce is ~70 us on one xGMI link, still far above the per-collective latency.
    The mean over ranks is ReduceOp.AVG inside the collective on RCCL (`nccl`), a per-bucket division after the wait elsewhere."""

    def __init__(self, module, bucket_cap_mb=None, process_group=None):
        super().__init__()
        if bucket_cap_mb is None:
            bucket_cap_mb = float(os.environ.get("Y5_DDP_BUCKET_MB", "6"))
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.avg_in_collective = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.params = list(module.parameters())
        with torch.no_grad():
            if self.world > 1:
                for t in list(module.parameters()) + [b for b in module.buffers() if b.dtype.is_floating_point or b.dtype == torch.long]:
                    dist.broadcast(t.data, 0, group=process_group)
        self.cap = int(bucket_cap_mb * 1024 * 1024 / 4)
        # CUs left to the collective while the backward plan runs (csrc/core.hip y5_set_cu_budget): the plan's persistent kernels size their grids for
        # CUs - reserve.  Default 0: swept on one rank through a real RCCL group (scripts/r5_ddp_reserve.py, profiles/r05/r05_ddp_exchange.log) a
        # reservation only slows the backward (exposed 0.60 / 0.62 / 0.73 / 0.90 / 1.64 ms at 0 / 8 / 16 / 32 / 64 CUs) -- a one-rank group launches NO
        # RCCL kernel (rocprofv3: 0 nccl dispatches), so its "exposed exchange" never was CU starvation.  What it is: see _launch.  The knob stays for
        # N > 1, where RCCL's ring kernels do need workgroup slots beside the plan's persistent workgroups.
        self.reserve_cus = int(os.environ.get("Y5_DDP_RESERVE_CUS", "0")) if self.avg_in_collective else 0
        self.buckets, self.p2b, self._eng, self._flat = [], {}, None, None
        module.__dict__["_ddp_sink"] = self
        for k in ("stride", "names", "hyp", "nc", "yaml"):
            if hasattr(module, k):
                setattr(self, k, getattr(module, k))

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def _attach(self, eng):
        """Cut the engine's arena (gradients in production order) into buckets of at most `cap` floats."""
        self._eng = eng
        self._flat = eng.be.view_torch(eng.gflat)
        order = sorted(eng.goff, key=eng.goff.get)
        self.buckets, cur, lo = [], [], 0
        for i in order:
            n = self.params[i].numel()
            end = eng.goff[i] + n
            if cur and end - lo > self.cap:
                self.buckets.append(_Bucket(cur, lo, eng.goff[i]))
                cur, lo = [], eng.goff[i]
            cur.append(i)
        if cur:
            self.buckets.append(_Bucket(cur, lo, eng.gtotal))
        self.p2b = {i: b for b in self.buckets for i in b.idxs}

    # ---- gradient sink protocol (called by TrainEngine.backward) -------------------------------------------------------
    def begin(self, eng):
        if eng is not self._eng:
            self._attach(eng)
        if self.reserve_cus > 0:
            _state.set_cu_budget(eng.lib, self._ncu(eng) - self.reserve_cus)
        for b in self.buckets:
            b.pending = len(b.idxs)
            b.work = None

    def _launch(self, b):
        if _libm.experimental("ddp_dry"):   # measurement aid (scripts/r5_ddp_reserve.py): bookkeeping only, no collective
            b.pending = -1
            return
        if self.world > 1 or self.avg_in_collective:  # (a 1-rank RCCL group still runs the collective: exercises the launch / wait path)
            op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
            # Y5_DDP_SYNC = none (default): asynchronous collectives on the process group's stream, overlapped with the rest of the backward plan (the
            # reference's DDP behaviour).  = all: issued synchronously, which this torch enqueues on the CURRENT stream -- no overlap, but no second
            # hardware queue either.  Measured through a one-rank group (profiles/r05/r05_ddp_exchange.log): bookkeeping alone 0 us, async collectives
            # 0.43-0.55 ms of step time whether 1 or 6 buckets (a fixed cost of bringing the second queue into play, not per collective and not the
            # last bucket's hand-over: `last` = 0.45-0.55 ms), synchronous 0.00-0.01 ms.  On N ranks the choice is that fixed cost against the un-overlapped
            # ring time of 28.9 MB (~0.3-0.5 ms estimated at 8 ranks): unmeasured here, so the overlapping form stays the default.
            mode = os.environ.get("Y5_DDP_SYNC", "none")   # none | last | all
            sync = self.avg_in_collective and (mode == "all" or (mode == "last" and b is self.buckets[-1]))
            w = dist.all_reduce(self._flat[b.lo:b.hi], op=op, group=self.pg, async_op=not sync)
            b.work = None if sync else w
        b.pending = -1

    def bucket_of(self, idx):
        """The bucket a parameter's gradient travels in (the engine batches its per-layer gradient unpacking by bucket)."""
        return self.p2b[idx]

    def grad_ready(self, idx):
        """The kernels producing parameter `idx`'s gradient are queued on the compute stream; a complete bucket goes on the
        wire at once (the collective is ordered after them by the process group's stream synchronisation)."""
        b = self.p2b[idx]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def abort(self):
        """The backward plan raised (out of memory, kernel error, skipped step): give the reserved CUs back -- the budget is process-global."""
        if self.reserve_cus > 0 and self._eng is not None:
            _state.set_cu_budget(self._eng.lib, 0)

    def finish(self, grads):
        """All kernels of the backward plan are queued: launch stragglers, wait for the wire, average."""
        for b in self.buckets:
            if b.pending > 0:  # parameters that received no gradient this step contribute zeros
                for i in b.idxs:
                    if grads[i] is None:
                        o = self._eng.goff[i]
                        self._flat[o:o + self.params[i].numel()].zero_()
                self._launch(b)
        if self.reserve_cus > 0:
            _state.set_cu_budget(self._eng.lib, 0)   # the forward / validation / optimizer launches that follow use every CU again
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()  # orders the compute stream behind the collective; no host block on RCCL
                if not self.avg_in_collective:
                    self._flat[b.lo:b.hi].div_(self.world)
        return grads

    @staticmethod
    def _ncu(eng):
        dev = getattr(eng.be, "device", None)
        return torch.cuda.get_device_properties(dev).multi_processor_count if dev is not None and torch.cuda.is_available() else 256


def scale_img(img, ratio=1.0, same_shape=False, gs=32, flip=None):
    """utils/torch_utils.py `scale_img` (+ the `x.flip(fi)` of models/yolo.py:276 folded in as `flip` = 2 | 3): bilinear resize of an NCHW float
    batch by `ratio` and padding with 0.447 to a multiple of `gs`, one `y5_scale_img` launch.  ratio == 1: the (flipped) image, unpadded, like
    the reference."""
    import ctypes as C

    from . import _lib
    if img.dtype not in (torch.float16, torch.float32) or img.dim() != 4 or not _lib.accepts(img):
        raise TypeError("scale_img: a float16 / float32 NCHW GPU batch is expected")
    img = img.contiguous()
    B, Cn, h, w = (int(v) for v in img.shape)
    if ratio == 1.0:
        s, (ph, pw) = (h, w), (h, w)
        if flip is None:
            return img
    else:
        s = (int(h * ratio), int(w * ratio))
        ph, pw = s if same_shape else tuple(math.ceil(v * ratio / gs) * gs for v in (h, w))
    out = torch.empty((B, Cn, ph, pw), dtype=img.dtype, device=img.device)
    code = _lib.Y5_F16 if img.dtype == torch.float16 else _lib.Y5_F32
    lib = _lib.lib()
    _lib.check(lib.y5_scale_img(C.c_void_p(img.data_ptr()), code, B, Cn, h, w, int(flip or 0), s[0], s[1], ph, pw, 0.447,
                                C.c_void_p(out.data_ptr()), code, _lib.stream(img.device)), lib)
    return out


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


def smart_optimizer(model, name="Adam", lr=0.001, momentum=0.9, decay=1e-5):
    """utils/torch_utils.py:257-290 (same defaults; train.py:227 passes every argument): 3 parameter groups -- biases (no decay),
    BatchNorm weights (no decay), other weights (decay).  name="SGD" returns the fused HipSGD."""
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
        # same groups / hyper-parameters / state_dict layout as torch.optim.SGD; the step is one fused multi-tensor launch
        optimizer = HipSGD(g[2], lr=lr, momentum=momentum, nesterov=True)
    else:
        raise NotImplementedError(f"Optimizer {name} not implemented.")
    optimizer.add_param_group({"params": g[0], "weight_decay": decay})
    optimizer.add_param_group({"params": g[1], "weight_decay": 0.0})
    return optimizer


class HipSGD(torch.optim.Optimizer):
    """SGD(momentum, nesterov, weight_decay) with torch.optim.SGD's constructor, param_groups and state (`momentum_buffer`), whose
    step is the fused multi-tensor kernel of csrc/optim.hip.  `step()` = plain optimizer step (gradients as they are);
    `step_fused(inv_scale, max_norm, ema, model)` folds train.py:413-421 into three launches: GradScaler unscale + inf check,
    clip_grad_norm_, the update (skipped on device when a gradient is non-finite) and ModelEMA.update.
    `lib` selects the kernel library (default: libyolov5_hip.so; the tests pass the host-compiled emulator for CPU tensors)."""

    def __init__(self, params, lr=0.01, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, lib=None):
        if dampening != 0.0:
            raise ValueError("HipSGD: dampening is not supported (the reference never sets it)")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay, nesterov=nesterov))
        self._lib = lib
        self._cache = None

    def _library(self):
        if self._lib is None:
            from . import _lib

            self._lib = _lib.lib()
        return self._lib

    def _tables(self, ema_pairs):
        """Device-resident y5_mt_tensor table (rebuilt only when a pointer changes: .grad tensors are views of the engine's
        gradient arena and keep their address from step to step)."""
        import ctypes as C

        import numpy as np

        from . import _lib

        if len(self.param_groups) > 4:
            raise ValueError("HipSGD: at most 4 parameter groups")
        rows = []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise TypeError("HipSGD: contiguous fp32 parameters and gradients only")
                st = self.state[p]
                if g["momentum"] != 0 and "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p)
                mom = st.get("momentum_buffer")
                e = ema_pairs.get(id(p)) if ema_pairs else None
                rows.append((p.data_ptr(), p.grad.data_ptr(), mom.data_ptr() if mom is not None else 0, e.data_ptr() if e is not None else 0,
                             p.numel(), gi))
        key = tuple(rows)
        if self._cache is None or self._cache[0] != key:
            if not rows:
                self._cache = (key, None, 0, 0, None, None)
                return self._cache
            dev = self.param_groups[0]["params"][0].device
            arr = (_lib.MtTensor * len(rows))()
            for r, (pp, gp, mp, ep, n, gi) in zip(arr, rows):
                r.param, r.grad, r.mom, r.ema, r.n, r.group = pp, gp, mp or None, ep or None, n, gi
            tab = torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy()).to(dev)
            max_n = max(r[4] for r in rows)
            ws = _lib.workspace(self._library().y5_mt_workspace_bytes(len(rows), max_n), dev)
            stats = torch.zeros(4, dtype=torch.float32, device=dev)
            self._cache = (key, tab, len(rows), max_n, ws, stats)
        return self._cache

    @staticmethod
    def _stream(dev):
        import ctypes as C

        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None

    @torch.no_grad()
    def step_fused(self, inv_scale=1.0, max_norm=0.0, ema=None, model=None):
        """One optimizer step; returns the device tensor [grad_norm, clip_coef, found_inf, -] when a norm pass ran, else None."""
        import ctypes as C

        from . import _lib

        if self._lib is None and any(not _lib.accepts(p) for g in self.param_groups for p in g["params"]):
            raise RuntimeError("HipSGD: parameters must live on the GPU (there is no CPU execution path)")
        lib = self._library()
        ema_pairs, d = None, 0.0
        if ema is not None:
            if model is None:
                raise ValueError("step_fused(ema=...) needs the model the EMA tracks")
            ema.updates += 1
            d = float(ema.decay(ema.updates))
            ema_pairs = ema.param_pairs(model)
        key, tab, n, max_n, ws, stats = self._tables(ema_pairs)
        if n == 0:
            return None
        dev = tab.device
        st = self._stream(dev)
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if g["momentum"] != g0["momentum"] or g["nesterov"] != g0["nesterov"]:
                raise ValueError("HipSGD: momentum / nesterov must be the same in every group")
        lr4 = (C.c_float * 4)(*([float(g["lr"]) for g in self.param_groups] + [0.0] * (4 - len(self.param_groups))))
        wd4 = (C.c_float * 4)(*([float(g["weight_decay"]) for g in self.param_groups] + [0.0] * (4 - len(self.param_groups))))
        use_stats = max_norm > 0 or inv_scale != 1.0
        if use_stats:
            _lib.check(lib.y5_mt_grad_norm(C.c_void_p(tab.data_ptr()), n, max_n, float(inv_scale), float(max_norm), C.c_void_p(stats.data_ptr()),
                                           C.c_void_p(ws.data_ptr()), ws.numel(), st), lib)
        _lib.check(lib.y5_mt_sgd_step(C.c_void_p(tab.data_ptr()), n, max_n, lr4, wd4, float(g0["momentum"]), int(bool(g0["nesterov"])),
                                      float(inv_scale), C.c_void_p(stats.data_ptr()) if use_stats else None, d, st), lib)
        if ema is not None:
            ema.lerp_rest(model, d, lib, st, skip=set(k[0] for k in key))
        _state.bump_weights_epoch()  # parameters (and the EMA's) were written through raw pointers: cached eval plans re-pack
        return stats if use_stats else None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_fused()
        return loss


class ModelEMA:
    """utils/torch_utils.py:343-375: EMA of parameters AND buffers, fp32, decay ramp 1 - exp(-updates / tau)."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = deepcopy(de_parallel(model)).eval()  # (BaseModel.__getstate__ leaves the engine caches / DDP sink behind)
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._pairs = None

    def update(self, model):
        self.updates += 1
        d = self.decay(self.updates)
        msd = de_parallel(model).state_dict()
        for k, v in self.ema.state_dict().items():
            if v.dtype.is_floating_point:
                v *= d
                v += (1 - d) * msd[k].detach()

    # ---- fused path (HipSGD.step_fused): the parameter EMA rides in the optimizer kernel, the rest in one y5_mt_lerp launch ----
    def _build_pairs(self, model):
        m = de_parallel(model)
        e_p, e_b = dict(self.ema.named_parameters()), dict(self.ema.named_buffers())
        params = {id(p): e_p[k] for k, p in m.named_parameters() if p.dtype.is_floating_point}
        rest = [(p, e_p[k]) for k, p in m.named_parameters() if p.dtype.is_floating_point]
        rest += [(b, e_b[k]) for k, b in m.named_buffers() if b.dtype.is_floating_point and k in e_b]
        # `nn.Module._apply` (`.half()`, `.float()`, `.to()`: val.py:187,388 does that to the EMA model once per epoch) REPLACES buffer
        # tensors while parameters keep their identity: remember where every cached buffer hangs so that a stale cache is noticed
        owners = []
        for root in (m, self.ema):
            for mod in root.modules():
                for name, b in mod._buffers.items():
                    if b is not None and b.dtype.is_floating_point:
                        owners.append((mod._buffers, name, b))
        self._pairs = (m, params, rest, {}, owners)
        return self._pairs

    def _current_pairs(self, model):
        pr = self._pairs
        if pr is None or pr[0] is not de_parallel(model) or any(d.get(k) is not t for d, k, t in pr[4]):
            pr = self._build_pairs(model)
        return pr

    def invalidate(self):
        """Drop the cached (model tensor, EMA tensor) pairs (after anything that re-creates tensors of either module)."""
        self._pairs = None

    def param_pairs(self, model):
        """id(model parameter) -> EMA parameter tensor."""
        return self._current_pairs(model)[1]

    def lerp_rest(self, model, d, lib, stream, skip=()):
        """EMA of every float tensor of the state_dict that the optimizer kernel did not already update (buffers: BatchNorm
        running statistics; parameters without a gradient)."""
        import ctypes as C

        import numpy as np

        from . import _lib

        pr = self._current_pairs(model)
        todo = [(s, e) for s, e in pr[2] if s.data_ptr() not in skip]
        if not todo:
            return
        key = tuple((s.data_ptr(), e.data_ptr()) for s, e in todo)
        tab = pr[3].get(key)
        if tab is None:
            pr[3].clear()
            arr = (_lib.MtTensor * len(todo))()
            for r, (s, e) in zip(arr, todo):
                if s.dtype != torch.float32 or e.dtype != torch.float32 or not s.is_contiguous() or not e.is_contiguous():
                    raise TypeError("ModelEMA fused update: contiguous fp32 tensors only")
                r.param, r.grad, r.mom, r.ema, r.n, r.group = s.data_ptr(), None, None, e.data_ptr(), s.numel(), 0
            tab = (torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy()).to(todo[0][0].device), max(s.numel() for s, _ in todo))
            pr[3][key] = tab
        _lib.check(lib.y5_mt_lerp(C.c_void_p(tab[0].data_ptr()), len(todo), tab[1], float(d), stream), lib)
        _state.bump_weights_epoch()
