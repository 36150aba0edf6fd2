"""`ComputeLoss` with the reference's constructor / call signature (utils/loss.py:101-247), executed by the HIP kernels
behind `y5_loss_forward` / `y5_loss_backward` (include/yolov5_hip.h).

    compute_loss = ComputeLoss(model)                 # model.hyp, model.model[-1] (Detect) are read like the reference
    loss, loss_items = compute_loss(p, targets)       # p: list of (bs, na, ny, nx, no) GPU tensors, targets (nt, 6)
    loss.backward()                                   # d loss / d p[i] from one fused kernel per level

No host synchronisation happens in either direction (the reference syncs on `if n := b.shape[0]`, loss.py:146), except with
`autobalance=True`, whose update rule (loss.py:173-177) is host arithmetic on each level's objectness loss: one read of nl floats per call.
Unsupported options raise: `sort_obj_iou`, `gr != 1`.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib


def smooth_bce(eps=0.1):
    """ultralytics.utils.metrics.smooth_bce as used at utils/loss.py:117."""
    return 1.0 - 0.5 * eps, 0.5 * eps


def de_parallel(model):
    return model.module if hasattr(model, "module") and isinstance(getattr(model, "module"), torch.nn.Module) and \
        type(model).__name__ in ("DistributedDataParallel", "DataParallel", "HipDDP") else model


def _void_pp(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, targets, *p):
        lib = _lib.lib()
        dev = p[0].device
        d, nt = owner._desc(p), int(targets.shape[0])
        nbytes = lib.y5_loss_workspace_bytes(C.byref(d), nt)
        if nbytes == 0:
            _lib.check(-1, lib)
        ws = _lib.workspace(nbytes, dev)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        st = _lib.stream(dev)
        p = [pi.contiguous() for pi in p]
        rc = lib.y5_loss_forward(C.byref(d), _void_pp(p), C.c_void_p(targets.data_ptr()) if nt else None, nt,
                                 C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, st)
        _lib.check(rc, lib)
        ctx.owner, ctx.d, ctx.nt, ctx.ws, ctx.nbytes, ctx.p = owner, d, nt, ws, nbytes, p
        owner._last = (d, nt, ws)
        loss, items = out[0:1], out[1:4]
        ctx.mark_non_differentiable(items)
        return loss, items

    @staticmethod
    def backward(ctx, g_loss, _g_items):
        lib = _lib.lib()
        p = ctx.p
        dev = p[0].device
        gs = g_loss.detach().to(torch.float32).reshape(-1)[:1].contiguous()
        dp = [torch.empty_like(pi) for pi in p]
        st = _lib.stream(dev)
        rc = lib.y5_loss_backward(C.byref(ctx.d), _void_pp(p), ctx.nt, C.c_void_p(gs.data_ptr()), _void_pp(dp),
                                  C.c_void_p(ctx.ws.data_ptr()), ctx.nbytes, st)
        _lib.check(rc, lib)
        return (None, None, *dp)


class ComputeLoss:
    """utils/loss.py:101-183."""

    sort_obj_iou = False
    # True (or env Y5_CHECK_TARGETS=1): validate targets[:, 0] < batch size and targets[:, 1] < nc on the host and raise IndexError
    # like the reference's indexing does (one device->host sync per call); the kernels themselves drop such rows (memory safe).
    check_targets = False

    def __init__(self, model, autobalance=False):
        h = model.hyp
        self.cp, self.cn = smooth_bce(eps=h.get("label_smoothing", 0.0))
        m = de_parallel(model).model[-1]  # Detect()
        self.balance = {3: [4.0, 1.0, 0.4]}.get(m.nl, [4.0, 1.0, 0.25, 0.06, 0.02])
        self.ssi = [float(v) for v in m.stride].index(16.0) if autobalance else 0  # loss.py:127 (ValueError without a stride-16 level, as there)
        self.gr, self.hyp, self.autobalance = 1.0, h, autobalance
        self.na, self.nc, self.nl = m.na, m.nc, m.nl
        self.anchors = m.anchors
        self.device = next(model.parameters()).device
        self._last = None
        if self.nl > 5 or self.na > 8:
            raise NotImplementedError("ComputeLoss supports up to 5 levels x 8 anchors")

    def _desc(self, p):
        if self.sort_obj_iou or self.gr != 1.0:
            raise NotImplementedError("sort_obj_iou / gr != 1 are not supported by the HIP loss kernels")
        dt = p[0].dtype
        if dt not in (torch.float16, torch.float32):
            raise TypeError(f"ComputeLoss: predictions must be float16 or float32, got {dt}")
        d = _lib.LossDesc()
        d.dtype = _lib.Y5_F16 if dt == torch.float16 else _lib.Y5_F32
        d.nl, d.na, d.nc, d.bs = self.nl, self.na, self.nc, int(p[0].shape[0])
        # host copy of the (static) anchors, refreshed only when the tensor was modified: a per-call .cpu() is a device->host
        # copy on the compute stream, i.e. a full pipeline drain in the middle of every training step
        ver = (self.anchors.data_ptr(), self.anchors._version)
        if getattr(self, "_anc_host", None) is None or self._anc_host[0] != ver:
            self._anc_host = (ver, self.anchors.detach().float().cpu())
        anc = self._anc_host[1]
        for i, pi in enumerate(p):
            if pi.dtype != dt or not _lib.accepts(pi):
                raise RuntimeError("ComputeLoss: every prediction level must be a GPU tensor of the same dtype (no CPU path)")
            if pi.dim() != 5 or pi.shape[1] != self.na or pi.shape[4] != 5 + self.nc:
                raise ValueError(f"ComputeLoss: level {i} has shape {tuple(pi.shape)}, expected (bs,{self.na},ny,nx,{5 + self.nc})")
            d.ny[i], d.nx[i] = int(pi.shape[2]), int(pi.shape[3])
            d.balance[i] = float(self.balance[i])
            for a in range(self.na):
                d.anchors[i * 16 + a * 2] = float(anc[i, a, 0])
                d.anchors[i * 16 + a * 2 + 1] = float(anc[i, a, 1])
        h = self.hyp
        d.hyp_box, d.hyp_obj, d.hyp_cls = float(h["box"]), float(h["obj"]), float(h["cls"])
        d.cls_pw, d.obj_pw, d.anchor_t = float(h["cls_pw"]), float(h["obj_pw"]), float(h["anchor_t"])
        d.fl_gamma = float(h.get("fl_gamma", 0.0))  # loss.py:120-122: > 0 wraps BCEcls and BCEobj in FocalLoss(gamma) (alpha = 0.25)
        d.cp, d.cn = float(self.cp), float(self.cn)
        return d

    def __call__(self, p, targets):
        if len(p) != self.nl:
            raise ValueError(f"ComputeLoss: expected {self.nl} prediction levels, got {len(p)}")
        targets = targets.to(device=p[0].device, dtype=torch.float32).contiguous()
        if (self.check_targets or os.environ.get("Y5_CHECK_TARGETS") == "1") and targets.numel():
            lo = targets[:, :2].min(0).values.tolist()
            hi = targets[:, :2].max(0).values.tolist()
            if lo[0] < 0 or hi[0] >= p[0].shape[0]:
                raise IndexError(f"ComputeLoss: target image index {int(hi[0] if hi[0] >= p[0].shape[0] else lo[0])} is out of bounds for batch size {p[0].shape[0]}")
            if lo[1] < 0 or hi[1] >= self.nc:
                raise IndexError(f"ComputeLoss: target class {int(hi[1] if hi[1] >= self.nc else lo[1])} is out of bounds for nc={self.nc}")
        out = _LossFn.apply(self, targets, *p)
        if self.autobalance:
            self._autobalance(len(p))
        return out

    def _autobalance(self, nl):
        """loss.py:173-177: balance[i] <- 0.9999 balance[i] + 0.0001 / obji (Python doubles on the fp32 obji), then / balance[ssi].  The
        forward that just ran (and its backward, which replays the descriptor captured there) used the OLD factors, like the reference's graph."""
        d, nt, ws = self._last
        off = _lib.lib().y5_loss_obji_offset(C.byref(d), nt)
        if off < 0:
            _lib.check(-1, _lib.lib())
        obji = ws[off:off + 4 * nl].view(torch.float32).tolist()  # the one device->host read of this option
        self.last_obji = obji
        for i in range(nl):
            self.balance[i] = self.balance[i] * 0.9999 + 0.0001 / obji[i]
        self.balance = [x / self.balance[self.ssi] for x in self.balance]

    def build_targets(self, p, targets):
        """utils/loss.py:185-247 -> (tcls, tbox, indices, anch) in the reference's format (int64 indices).
        Runs the forward kernels and slices the rows out of the workspace (one host read of the row counts)."""
        self.__call__([pi.detach() for pi in p], targets)
        d, nt, ws = self._last
        lib = _lib.lib()
        tcls, tbox, indices, anch = [], [], [], []
        offs = (C.c_size_t * 10)()
        cap = C.c_longlong(0)
        for i in range(self.nl):
            _lib.check(lib.y5_loss_targets_layout(C.byref(d), nt, i, offs, C.byref(cap)), lib)
            n = int(ws[offs[0]:offs[0] + 4].view(torch.int32)[0])

            def arr(k, dtype, width):
                es = 4
                return ws[offs[k]:offs[k] + n * width * es].view(dtype).reshape(n, width) if width > 1 else \
                    ws[offs[k]:offs[k] + n * es].view(dtype)

            b, a, gj, gi, c = (arr(k, torch.int32, 1).long() for k in (1, 2, 3, 4, 5))
            indices.append((b, a, gj, gi))
            tcls.append(c)
            tbox.append(arr(6, torch.float32, 4).clone())
            anch.append(arr(7, torch.float32, 2).clone())
        return tcls, tbox, indices, anch
