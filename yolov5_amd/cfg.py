"""Built-in architecture configs (same schema as the reference's models/*.yaml: nc, depth_multiple,
width_multiple, anchors, backbone, head) for yolov5{n,s,m,l,x} and the -seg variants, plus the default
training hyper-parameters (data/hyps/hyp.scratch-low.yaml) used by ComputeLoss.

A user-supplied *.yaml path or dict is accepted everywhere a config name is (models/yolo.py:218-231 behaviour).
"""
from __future__ import annotations

import copy
import os

_MULT = {"n": (0.33, 0.25), "s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}
_ANCHORS = ((10, 13, 16, 30, 33, 23), (30, 61, 62, 45, 59, 119), (116, 90, 156, 198, 373, 326))


def builtin_cfg(name: str) -> dict:
    """name: 'yolov5s', 'yolov5s.yaml', 'yolov5x-seg.yaml', ... -> model dict."""
    stem = os.path.basename(name).replace(".yaml", "")
    seg = stem.endswith("-seg")
    size = stem.replace("-seg", "")[-1:]
    if not stem.startswith("yolov5") or size not in _MULT:
        raise FileNotFoundError(f"unknown model config '{name}' (built-ins: yolov5[nsmlx][-seg])")
    gd, gw = _MULT[size]
    backbone = [
        [-1, 1, "Conv", [64, 6, 2, 2]],  # 0-P1/2
        [-1, 1, "Conv", [128, 3, 2]],  # 1-P2/4
        [-1, 3, "C3", [128]],
        [-1, 1, "Conv", [256, 3, 2]],  # 3-P3/8
        [-1, 6, "C3", [256]],
        [-1, 1, "Conv", [512, 3, 2]],  # 5-P4/16
        [-1, 9, "C3", [512]],
        [-1, 1, "Conv", [1024, 3, 2]],  # 7-P5/32
        [-1, 3, "C3", [1024]],
        [-1, 1, "SPPF", [1024, 5]],  # 9
    ]
    head = [
        [-1, 1, "Conv", [512, 1, 1]],
        [-1, 1, "nn.Upsample", ["None", 2, "nearest"]],
        [[-1, 6], 1, "Concat", [1]],  # cat backbone P4
        [-1, 3, "C3", [512, False]],  # 13
        [-1, 1, "Conv", [256, 1, 1]],
        [-1, 1, "nn.Upsample", ["None", 2, "nearest"]],
        [[-1, 4], 1, "Concat", [1]],  # cat backbone P3
        [-1, 3, "C3", [256, False]],  # 17 (P3/8-small)
        [-1, 1, "Conv", [256, 3, 2]],
        [[-1, 14], 1, "Concat", [1]],  # cat head P4
        [-1, 3, "C3", [512, False]],  # 20 (P4/16-medium)
        [-1, 1, "Conv", [512, 3, 2]],
        [[-1, 10], 1, "Concat", [1]],  # cat head P5
        [-1, 3, "C3", [1024, False]],  # 23 (P5/32-large)
        [[17, 20, 23], 1, "Segment", ["nc", "anchors", 32, 256]] if seg else [[17, 20, 23], 1, "Detect", ["nc", "anchors"]],
    ]
    return {"nc": 80, "depth_multiple": gd, "width_multiple": gw, "anchors": [list(a) for a in _ANCHORS],
            "backbone": backbone, "head": head}


def load_cfg(cfg) -> dict:
    """dict -> deep copy; existing *.yaml path -> parsed; otherwise a built-in name."""
    if isinstance(cfg, dict):
        return copy.deepcopy(cfg)
    cfg = str(cfg)
    if os.path.isfile(cfg):
        import yaml

        with open(cfg, encoding="ascii", errors="ignore") as f:
            return yaml.safe_load(f)
    return builtin_cfg(cfg)


# data/hyps/hyp.scratch-low.yaml (the reference's default --hyp, train.py:562)
HYP_SCRATCH_LOW = {
    "lr0": 0.01, "lrf": 0.01, "momentum": 0.937, "weight_decay": 0.0005, "warmup_epochs": 3.0,
    "warmup_momentum": 0.8, "warmup_bias_lr": 0.1, "box": 0.05, "cls": 0.5, "cls_pw": 1.0, "obj": 1.0,
    "obj_pw": 1.0, "iou_t": 0.20, "anchor_t": 4.0, "fl_gamma": 0.0, "hsv_h": 0.015, "hsv_s": 0.7, "hsv_v": 0.4,
    "degrees": 0.0, "translate": 0.1, "scale": 0.5, "shear": 0.0, "perspective": 0.0, "flipud": 0.0,
    "fliplr": 0.5, "mosaic": 1.0, "mixup": 0.0, "copy_paste": 0.0,
}
