"""Model classes with the reference's names and API (models/yolo.py): Detect :71-128, Segment :131-150,
BaseModel :153-212, DetectionModel :215-327, SegmentationModel :333, parse_model :375-458.

Eval-mode `forward` is one call into the HIP execution plan (yolov5_amd.engine); what it returns matches the
reference: `(z[bs, N, no], [raw_i[bs, na, ny, nx, no]])`, `(z,)` when `Detect.export` is set (AutoShape), and
`(z, proto, raw)` / `(z, proto)` for Segment.  Training-mode forward returns the raw maps and is differentiable
w.r.t. the parameters through the HIP backward plan of yolov5_amd/train_engine.py.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from copy import deepcopy
from pathlib import Path

import torch
from torch import nn

from . import _lib, _state
from .cfg import load_cfg
from .common import C3, SPPF, Bottleneck, Concat, Conv, Proto
from .general import LOGGER, make_divisible
from .packing import fuse_conv_bn_weights


class Detect(nn.Module):
    """Detect head (models/yolo.py:71-128).  The 1x1 convs, sigmoid, grid/anchor decode and the (bs,N,no) concat
    run as y5_conv2d_fwd + y5_detect_decode inside the model plan."""

    stride = None
    dynamic = False
    export = False

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.empty(0) for _ in range(self.nl)]
        self.anchor_grid = [torch.empty(0) for _ in range(self.nl)]
        self.register_buffer("anchors", torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.inplace = inplace

    def forward(self, x):
        raise RuntimeError("yolov5_amd.Detect is executed by the model engine; call the parent DetectionModel")

    def _make_grid(self, nx=20, ny=20, i=0):
        """Reference-identical grid / anchor_grid tensors (models/yolo.py:117-128); the kernel computes them on the
        fly, this helper exists for API parity and for the bit-exactness test."""
        d, t = self.anchors[i].device, self.anchors[i].dtype
        shape = 1, self.na, ny, nx, 2
        y, x = torch.arange(ny, device=d, dtype=t), torch.arange(nx, device=d, dtype=t)
        yv, xv = torch.meshgrid(y, x, indexing="ij")
        grid = torch.stack((xv, yv), 2).expand(shape) - 0.5
        anchor_grid = (self.anchors[i] * self.stride[i]).view((1, self.na, 1, 1, 2)).expand(shape)
        return grid, anchor_grid


class Segment(Detect):
    """Segment head (models/yolo.py:131-150): Detect with nm mask coefficients + Proto."""

    def __init__(self, nc=80, anchors=(), nm=32, npr=256, ch=(), inplace=True):
        super().__init__(nc, anchors, ch, inplace)
        self.nm = nm
        self.npr = npr
        self.no = 5 + nc + self.nm
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.proto = Proto(ch[0], self.npr, self.nm)


class BaseModel(nn.Module):
    """models/yolo.py:153-212."""

    def forward(self, x, profile=False):
        return self._forward_once(x, profile)

    def _forward_once(self, x, profile=False):
        if self.training:
            from .train_engine import train_forward

            return train_forward(self, x)  # list of raw (bs, na, ny, nx, no) maps, differentiable w.r.t. the parameters
        det = self.model[-1]
        want_raw = not getattr(det, "export", False)
        key = (tuple(x.shape), next(self.parameters()).dtype, str(x.device), want_raw)
        cache = self._engines
        eng = cache.get(key)
        stamp = (_state.weights_epoch, self._weights_version())
        if eng is None:
            from .engine import Engine, SplitEngine

            parts = 2 if _lib.experimental("split2") else 1
            if x.is_cuda and parts > 1 and x.shape[0] >= 16 * parts and x.shape[0] % parts == 0:
                # opt-in (Y5_EXPERIMENTAL=split2): sub-batch plans on separate streams fill each other's kernel tails (engine.SplitEngine;
                # +3 % images/s on yolov5s bs=64, but per-kernel figures then describe overlapped launches)
                eng = SplitEngine(self, tuple(x.shape), key[1], x.device, want_raw=want_raw, parts=parts)
            else:
                eng = Engine(self, tuple(x.shape), key[1], x.device, want_raw=want_raw)
            eng._stamp = stamp
            cache[key] = eng
            # a few live plans per model (rectangular validation batches, val.py rect=True, alternate between shapes); the least
            # recently used one is dropped -- a yolov5s bs=64 640^2 plan holds 2.6 GB of the 288 GB
            while len(cache) > max(1, int(os.environ.get("Y5_PLAN_CACHE", "4"))):
                cache.popitem(last=False)
        else:
            cache.move_to_end(key)
            if eng._stamp != stamp:  # weights / BN statistics changed since the filters were packed: re-pack, keep plan + graphs
                eng.refresh_weights()
                eng._stamp = stamp
        out = eng(x)
        z = out["z"]
        raw = [out[f"raw{i}"] for i in range(det.nl)] if want_raw else None
        if isinstance(det, Segment):
            return (z, out["proto"]) if not want_raw else (z, out["proto"], raw)
        return (z,) if not want_raw else (z, raw)

    @property
    def _engines(self):
        e = self.__dict__.get("_engine_cache")
        if e is None:
            e = self.__dict__["_engine_cache"] = OrderedDict()
        return e

    def _weights_version(self):
        """Sum of torch's in-place version counters over parameters and buffers (tensor list cached until the module tree changes:
        fuse() / _apply() call invalidate_engine).  Catches load_state_dict / optimizer.step / user edits made through torch;
        raw-pointer writers bump _state.weights_epoch instead."""
        ts = self.__dict__.get("_ver_tensors")
        if ts is None:
            ts = self.__dict__["_ver_tensors"] = list(self.parameters()) + list(self.buffers())
        return sum(t._version for t in ts)

    def invalidate_engine(self):
        """Drop the cached plans (module tree / dtype / device changed)."""
        self._engines.clear()
        self.__dict__.pop("_ver_tensors", None)

    def __getstate__(self):
        """deepcopy / pickle (ModelEMA, train.py:469-488 checkpoints): engines hold ctypes handles and device plans -- they are
        per-process caches, not model state."""
        st = self.__dict__.copy()
        for k in ("_engine_cache", "_train_engines", "_ddp_sink", "_ver_tensors"):
            st.pop(k, None)
        return st

    def fuse(self):
        """Fold BN into conv in every Conv block (models/yolo.py:186-195, utils/torch_utils.py:224-254)."""
        LOGGER.info("Fusing layers... ")
        for m in self.model.modules():
            if isinstance(m, Conv) and hasattr(m, "bn"):
                cv, bn = m.conv, m.bn
                w, b = fuse_conv_bn_weights(cv.weight.detach(), None if cv.bias is None else cv.bias.detach(), bn.weight.detach(),
                                            bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
                fused = nn.Conv2d(cv.in_channels, cv.out_channels, cv.kernel_size, cv.stride, cv.padding, bias=True)
                fused = fused.requires_grad_(False).to(cv.weight.device)
                fused.weight.copy_(w.to(cv.weight.dtype))
                fused.bias.copy_(b.to(cv.weight.dtype))
                m.conv = fused
                delattr(m, "bn")
        self.invalidate_engine()
        return self

    def info(self, verbose=False, img_size=640):
        n_p = sum(x.numel() for x in self.parameters())
        n_l = len(list(self.modules()))
        LOGGER.info(f"{type(self).__name__} summary: {n_l} layers, {n_p} parameters")

    def _apply(self, fn):
        """Also move Detect's non-buffer tensors (models/yolo.py:201-212) and drop the cached plan."""
        super()._apply(fn)
        m = self.model[-1]
        if isinstance(m, Detect):
            if m.stride is not None:
                m.stride = fn(m.stride)
            m.grid = list(map(fn, m.grid))
            if isinstance(m.anchor_grid, list):
                m.anchor_grid = list(map(fn, m.anchor_grid))
        self.invalidate_engine()
        return self


def check_anchor_order(m):
    """utils/autoanchor.py:16-23."""
    a = m.anchors.prod(-1).mean(-1).view(-1)
    da = a[-1] - a[0]
    ds = m.stride[-1] - m.stride[0]
    if da and (da.sign() != ds.sign()):
        LOGGER.info("AutoAnchor: Reversing anchor order")
        m.anchors[:] = m.anchors.flip(0)


def _graph_strides(layers, save_from):
    """Stride of every Detect input by shape inference over the layer list.  The reference runs a 256x256 dummy
    forward instead (models/yolo.py:250-256); the result is the same integer down-sampling factors."""
    scale = []
    for i, m in enumerate(layers):
        f = m.f
        srcs = [f] if isinstance(f, int) else list(f)
        srcs = [i - 1 if j == -1 else j for j in srcs]
        s_in = 1.0 if i == 0 else scale[srcs[0]] if not isinstance(m, Detect) else None
        if isinstance(m, Conv):
            st = m.conv.stride
            scale.append(s_in * (st[0] if isinstance(st, tuple) else st))
        elif isinstance(m, nn.Upsample):
            scale.append(s_in / float(m.scale_factor))
        elif isinstance(m, Detect):
            return [float(scale[j]) for j in srcs]
        else:
            scale.append(s_in)
    return []


class DetectionModel(BaseModel):
    """models/yolo.py:215-327."""

    def __init__(self, cfg="yolov5s.yaml", ch=3, nc=None, anchors=None):
        super().__init__()
        self.yaml = load_cfg(cfg)
        if not isinstance(cfg, dict):
            self.yaml_file = Path(str(cfg)).name
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            LOGGER.info(f"Overriding model.yaml nc={self.yaml['nc']} with nc={nc}")
            self.yaml["nc"] = nc
        if anchors:
            LOGGER.info(f"Overriding model.yaml anchors with anchors={anchors}")
            self.yaml["anchors"] = round(anchors)
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.names = [str(i) for i in range(self.yaml["nc"])]
        self.inplace = self.yaml.get("inplace", True)

        m = self.model[-1]
        if isinstance(m, Detect):
            m.inplace = self.inplace
            m.stride = torch.tensor(_graph_strides(list(self.model), self.save))
            check_anchor_order(m)
            m.anchors /= m.stride.view(-1, 1, 1)  # yolo.py:254
            self.stride = m.stride
            self._initialize_biases()
        initialize_weights(self)

    def forward(self, x, augment=False, profile=False):
        if augment:
            return self._forward_augment(x)
        return self._forward_once(x, profile)

    def _forward_augment(self, x):
        """models/yolo.py:269-282: three forwards (scales 1 / 0.83 / 0.67, the middle one left-right flipped), predictions de-scaled / de-flipped
        (`_descale_pred`, :283-299, one in-place launch each) and concatenated without the overlapping tails (`_clip_augmented`, :301-312)."""
        import ctypes as C

        from . import _lib
        from .torch_utils import scale_img
        if self.training:
            raise RuntimeError("augmented inference runs in eval mode")
        img_size = x.shape[-2:]
        gs = int(self.stride.max())
        lib, y = _lib.lib(), []
        for si, fi in zip((1, 0.83, 0.67), (None, 3, None)):
            xi = scale_img(x, si, gs=gs, flip=fi)
            yi = self._forward_once(xi)[0].clone()      # (the plan owns its z; the de-scaling below is in place)
            code = _lib.Y5_F16 if yi.dtype == torch.float16 else _lib.Y5_F32
            _lib.check(lib.y5_tta_descale(C.c_void_p(yi.data_ptr()), code, yi.shape[0] * yi.shape[1], yi.shape[2], float(si), int(fi or 0),
                                          float(img_size[0]), float(img_size[1]), _lib.stream(yi.device)), lib)
            y.append(yi)
        y = self._clip_augmented(y)
        return torch.cat(y, 1), None

    def _clip_augmented(self, y):
        """models/yolo.py:301-312."""
        nl = self.model[-1].nl
        g = sum(4 ** k for k in range(nl))
        e = 1
        i = (y[0].shape[1] // g) * sum(4 ** k for k in range(e))
        y[0] = y[0][:, :-i]
        i = (y[-1].shape[1] // g) * sum(4 ** (nl - 1 - k) for k in range(e))
        y[-1] = y[-1][:, i:]
        return y

    def _initialize_biases(self, cf=None):
        """models/yolo.py:314-327."""
        m = self.model[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:5 + m.nc] += math.log(0.6 / (m.nc - 0.99999)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)


Model = DetectionModel  # models/yolo.py:330


class SegmentationModel(DetectionModel):
    """models/yolo.py:333-341."""

    def __init__(self, cfg="yolov5s-seg.yaml", ch=3, nc=None, anchors=None):
        super().__init__(cfg, ch, nc, anchors)


def initialize_weights(model):
    """ultralytics initialize_weights as called at models/yolo.py:259: BN eps=1e-3, momentum=0.03."""
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif type(m) in {nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU}:
            m.inplace = True


_MODULES = {"Conv": Conv, "C3": C3, "SPPF": SPPF, "Bottleneck": Bottleneck, "Concat": Concat, "Detect": Detect,
            "Segment": Segment, "nn.Upsample": nn.Upsample, "Proto": Proto}


def parse_model(d, ch):
    """models/yolo.py:375-458 restricted to the module set of the shipped yolov5{n,s,m,l,x}[-seg] configs."""
    anchors, nc, gd, gw = d["anchors"], d["nc"], d["depth_multiple"], d["width_multiple"]
    if d.get("activation"):
        raise NotImplementedError("custom activations are not supported by the fused HIP epilogue")
    ch_mul = d.get("channel_multiple") or 8
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        if isinstance(m, str):
            if m not in _MODULES:
                raise NotImplementedError(f"module '{m}' is not part of the YOLOv5 hot path (SURVEY 2: out of scope)")
            mname, m = m, _MODULES[m]
        else:
            mname = m.__name__
        args = list(args)
        for j, a in enumerate(args):
            if isinstance(a, str):
                args[j] = {"nc": nc, "anchors": anchors, "None": None, "False": False, "True": True}.get(a, a)
        n = n_ = max(round(n * gd), 1) if n > 1 else n
        if m in {Conv, Bottleneck, SPPF, C3}:
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, ch_mul)
            args = [c1, c2, *args[1:]]
            if m is C3:
                args.insert(2, n)
                n = 1
        elif m is Concat:
            c2 = sum(ch[x] for x in f)
        elif m in {Detect, Segment}:
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
            if m is Segment:
                args[3] = make_divisible(args[3] * gw, ch_mul)
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        t = f"models.common.{mname}" if m not in {Detect, Segment} else f"models.yolo.{mname}"
        np_ = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type, m_.np = i, f, t, np_
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)
