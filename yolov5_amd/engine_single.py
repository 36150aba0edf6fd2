"""Standalone forward of one layer (Conv / Bottleneck / C3 / SPPF / Proto): NCHW in -> NCHW out, executed by the same
HIP kernels and planner as the full model.  Exists for API parity with the reference's nn.Module layers
(models/common.py) and for per-layer parity tests; the hot path is the whole-model plan."""
from __future__ import annotations

import torch

from . import _lib
from .engine import Engine, PlanSpec, _Planner


class _SinglePlanner(_Planner):
    def __init__(self, layer, kind, B, ch, H, W):
        from . import common, yolo

        self.cm, self.yo = common, yolo
        self.layer, self.kind = layer, kind
        self.spec = PlanSpec(B, ch, (H, W))
        self.want_raw = False

    def run(self):
        spec, m = self.spec, self.layer
        H, W = spec.in_hw
        cpad = (spec.in_ch + 7) // 8 * 8
        x0 = spec.new_buf(H, W, cpad, "input_nhwc")
        spec.ops.append(dict(op="to_nhwc", dst=x0, C=spec.in_ch))
        x = x0 if cpad == spec.in_ch else x0  # padded channels are zero and meet zero filter taps
        if self.kind == "conv":
            y = self.conv([m], x, None, name="Conv", view="first")
        elif self.kind == "bottleneck":
            tmp = spec.new_buf(H, W, m.cv1.conv.out_channels, "tmp")
            t = self.conv([m.cv1], x, tmp, name="b.cv1", view="first")
            y = self.conv([m.cv2], t, None, res=x if m.add else None, name="b.cv2")
        elif self.kind == "c3":
            y = self.c3_first(m, x)
        elif self.kind == "sppf":
            y = self.sppf_first(m, x)
        elif self.kind == "proto":
            y = self.proto_first(m, x)
        else:
            raise ValueError(self.kind)
        spec.ops.append(dict(op="to_nchw", src=y, out="y"))
        spec.outputs["y"] = dict(shape=(spec.B, y.C, y.H, y.W))
        return spec

    # variants whose first conv reads the channel-padded input buffer
    def c3_first(self, m, x):
        from .engine import _slice

        c_ = m.cv1.conv.out_channels
        cat = self.spec.new_buf(x.H, x.W, 2 * c_, "C3.cat")
        self.conv([m.cv1, m.cv2], x, cat, name="C3.cv1+cv2", view="first")
        a = _slice(cat, 0, c_)
        if len(m.m):
            tmp = self.spec.new_buf(x.H, x.W, c_, "C3.tmp")
            for b in m.m:
                self.bottleneck(b, a, a, tmp)
        return self.conv([m.cv3], cat, None, name="C3.cv3")

    def sppf_first(self, m, x):
        from .engine import _slice

        c_ = m.cv1.conv.out_channels
        k = m.m.kernel_size if isinstance(m.m.kernel_size, int) else m.m.kernel_size[0]
        cat = self.spec.new_buf(x.H, x.W, 4 * c_, "SPPF.cat")
        self.conv([m.cv1], x, _slice(cat, 0, c_), name="SPPF.cv1", view="first")
        self.spec.ops.append(dict(op="sppf_pool", buf=cat, C=c_, k=k))
        return self.conv([m.cv2], cat, None, name="SPPF.cv2")

    def proto_first(self, m, x):
        c_ = m.cv1.conv.out_channels
        up = self.spec.new_buf(2 * x.H, 2 * x.W, c_, "proto.up")
        self.conv([m.cv1], x, None, up2=up, name="proto.cv1", view="first")
        t = self.conv([m.cv2], up, None, name="proto.cv2")
        return self.conv([m.cv3], t, None, name="proto.cv3")


class _SingleEngine(Engine):
    def __init__(self, layer, kind, x_shape, dtype, device, backend=None):
        B, ch, H, W = x_shape
        spec = _SinglePlanner(layer, kind, B, ch, H, W).run()
        super().__init__(None, x_shape, dtype, device, backend=backend, spec=spec)


def run_single(layer, x, kind):
    if layer.training:
        raise NotImplementedError("yolov5_amd: training-mode layer forward is not built yet; call .eval()")
    if not _lib.accepts(x):
        raise RuntimeError("yolov5_amd: input tensor must live on the GPU (no CPU path)")
    dtype = next(layer.parameters()).dtype
    key = (kind, tuple(x.shape), dtype, str(x.device), sum(p._version for p in layer.parameters()))
    cache = layer.__dict__.setdefault("_single_engine", {})
    eng = cache.get(key)
    if eng is None:
        cache.clear()
        eng = cache[key] = _SingleEngine(layer, kind, tuple(x.shape), dtype, x.device)
    return eng(x.to(dtype))["y"].clone()
