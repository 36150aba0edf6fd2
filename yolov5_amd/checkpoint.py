"""Checkpoint round trip in the reference's format (train.py:469-488 save, utils/torch_utils.py:293-312 `smart_resume`,
models/experimental.py:60-101 `attempt_load`).

A reference checkpoint is a pickled dict {"epoch", "best_fitness", "model": fp16 nn.Module, "ema", "updates", "optimizer", "opt",
"git", "date"} whose module classes are found by PATH (`models.yolo.DetectionModel`, `models.common.Conv`, ...).  The classes of
yolov5_amd.yolo / yolov5_amd.common have the reference's attribute layout, so the two directions work by mapping paths:
  * loading:  experimental.install_reference_aliases() registers `models.yolo` / `models.common` modules whose names resolve to the
    yolov5_amd classes -- official *.pt files unpickle straight into this engine;
  * saving:   `save_checkpoint(..., reference_paths=True)` writes the yolov5_amd classes UNDER those reference paths, so the file
    loads in a stock ultralytics/yolov5 checkout (its own classes, its own forward) -- verified in tests/test_checkpoint.py by
    unpickling with the unmodified reference and comparing outputs.
Engine caches (ctypes handles, device plans) never enter a pickle (BaseModel.__getstate__)."""
from __future__ import annotations

import contextlib
import sys
from copy import deepcopy
from datetime import datetime

import torch
from torch import nn

from .torch_utils import de_parallel


@contextlib.contextmanager
def reference_class_paths():
    """Inside: every class of yolov5_amd.yolo / yolov5_amd.common pickles as `models.yolo.<Name>` / `models.common.<Name>`."""
    from . import common, experimental, yolo

    saved_mods = {k: sys.modules.get(k) for k in ("models", "models.yolo", "models.common", "models.experimental")}
    for k in saved_mods:
        sys.modules.pop(k, None)
    experimental.install_reference_aliases()
    touched = []
    try:
        for src, path in ((yolo, "models.yolo"), (common, "models.common")):
            for v in vars(src).values():
                if isinstance(v, type) and v.__module__ == src.__name__:
                    touched.append((v, v.__module__))
                    v.__module__ = path
        yield
    finally:
        for cls, mod in touched:
            cls.__module__ = mod
        for k, v in saved_mods.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def save_checkpoint(path, model, ema=None, optimizer=None, epoch=-1, best_fitness=None, opt=None, reference_paths=True):
    """train.py:469-488: model / EMA as fp16 deep copies, optimizer state_dict (HipSGD keeps torch.optim.SGD's layout), bookkeeping."""
    ckpt = {
        "epoch": epoch,
        "best_fitness": best_fitness,
        "model": deepcopy(de_parallel(model)).half(),
        "ema": deepcopy(ema.ema).half() if ema is not None else None,
        "updates": ema.updates if ema is not None else 0,
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "opt": dict(opt) if opt is not None else None,
        "git": None,
        "date": datetime.now().isoformat(),
    }
    for k in ("model", "ema"):  # plain CPU pickles, like the reference's (map_location handles the rest)
        if ckpt[k] is not None:
            ckpt[k] = ckpt[k].cpu()
    with (reference_class_paths() if reference_paths else contextlib.nullcontext()):
        torch.save(ckpt, str(path))
    return ckpt


def load_checkpoint(path, map_location="cpu"):
    """torch.load of a reference-format checkpoint (either class-path flavour) -> dict."""
    from .experimental import install_reference_aliases

    install_reference_aliases()
    return torch.load(str(path), map_location=map_location, weights_only=False)


def smart_resume(ckpt, optimizer, ema=None, weights="yolov5s.pt", epochs=300, resume=True):
    """utils/torch_utils.py:293-312."""
    best_fitness = 0.0
    start_epoch = ckpt["epoch"] + 1
    if ckpt["optimizer"] is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
        best_fitness = ckpt["best_fitness"]
    if ema and ckpt.get("ema"):
        ema.ema.load_state_dict(ckpt["ema"].float().state_dict())
        ema.updates = ckpt["updates"]
    if resume:
        assert start_epoch > 0, f"{weights} training to {epochs} epochs is finished, nothing to resume."
    if epochs < start_epoch:
        epochs += ckpt["epoch"]
    return best_fitness, start_epoch, epochs


def strip_optimizer(f="best.pt", s="", reference_paths=True):
    """utils/general.py:770-787: finalise a training checkpoint -- the EMA becomes the model, the optimizer / EMA / bookkeeping entries
    are cleared, epoch = -1, weights fp16 with requires_grad off; written to `s` or back over `f`.  Returns the file size in MB."""
    import os

    x = load_checkpoint(f)
    if x.get("ema"):
        x["model"] = x["ema"]
    for k in ("optimizer", "best_fitness", "ema", "updates"):
        x[k] = None
    x["epoch"] = -1
    x["model"].half()
    for p in x["model"].parameters():
        p.requires_grad = False
    with (reference_class_paths() if reference_paths else contextlib.nullcontext()):
        torch.save(x, str(s or f))
    return os.path.getsize(s or f) / 1e6
