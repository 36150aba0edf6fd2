"""Checkpoint loading (models/experimental.py:60-101 `attempt_load`): the reference stores PICKLED nn.Modules whose
class paths are `models.yolo.DetectionModel`, `models.common.Conv`, ...  `install_reference_aliases()` maps those
module paths onto the yolov5_amd classes (same attribute layout), so official *.pt files unpickle into this engine.
"""
from __future__ import annotations

import sys
import types

import torch
from torch import nn


def install_reference_aliases():
    from . import common, general, yolo

    if "models.yolo" in sys.modules and getattr(sys.modules["models.yolo"], "__y5amd__", False):
        return
    if "models.yolo" in sys.modules:
        # a REAL ultralytics/yolov5 checkout is imported in this process: its classes own those paths (pickles resolve to them)
        raise RuntimeError("yolov5_amd: the reference's `models` package is already imported; cannot alias its class paths")
    pk = types.ModuleType("models")
    pk.__path__ = []
    my = types.ModuleType("models.yolo")
    mc = types.ModuleType("models.common")
    me = types.ModuleType("models.experimental")
    for src, dst in ((yolo, my), (common, mc)):
        for k, v in vars(src).items():
            if isinstance(v, type):
                setattr(dst, k, v)
    my.__y5amd__ = True
    pk.yolo, pk.common, pk.experimental = my, mc, me
    sys.modules.setdefault("models", pk)
    sys.modules["models.yolo"] = my
    sys.modules["models.common"] = mc
    sys.modules["models.experimental"] = me


def attempt_load(weights, device=None, inplace=True, fuse=True):
    """Load one checkpoint (ensembles are out of scope) -> fused, eval-mode model (experimental.py:69-101)."""
    install_reference_aliases()
    w = weights[0] if isinstance(weights, (list, tuple)) else weights
    ckpt = torch.load(str(w), map_location="cpu", weights_only=False)
    model = ckpt.get("ema") or ckpt["model"] if isinstance(ckpt, dict) else ckpt
    model = model.float()
    if not hasattr(model, "stride"):
        model.stride = torch.tensor([32.0])
    if hasattr(model, "names") and isinstance(model.names, (list, tuple)):
        model.names = dict(enumerate(model.names))
    model = (model.fuse() if fuse and hasattr(model, "fuse") else model).eval()
    for m in model.modules():
        if type(m) in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU):
            m.inplace = inplace
        elif isinstance(m, nn.Upsample) and not hasattr(m, "recompute_scale_factor"):
            m.recompute_scale_factor = None
    return model.to(device) if device is not None else model
