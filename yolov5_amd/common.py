"""Layer classes with the reference's names, constructor signatures and attribute layout (models/common.py), so
that YAML configs, `state_dict`s and pickled checkpoints of the reference map 1:1 -- but `forward` runs the
hand-written HIP kernels through yolov5_amd.engine (no torch ops, no CPU path).

Covered: autopad :62, Conv :74-92, Bottleneck :164-181, C3 :230-246, SPPF :318-340, Concat :443-453,
Proto :1104-1117, DetectMultiBackend (pt branch) :456-814, AutoShape :843-948, Detections (tensor part) :950-1101.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn


def autopad(k, p=None, d=1):
    """'same' padding (models/common.py:62-71)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class _HipModule(nn.Module):
    """Mixin: standalone `forward` of a layer = a one-layer engine (same kernels as the full model)."""

    def _run_single(self, x, kind):
        from .engine_single import run_single

        return run_single(self, x, kind)


class Conv(_HipModule):
    """conv -> BN -> SiLU (models/common.py:74-92).  After `fuse()` the BN is folded and `bn` is deleted."""

    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1:
            raise NotImplementedError("yolov5_amd.Conv: groups/dilation != 1 are outside the hot path (SURVEY 2a)")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        if not isinstance(self.act, (nn.SiLU, nn.Identity)):
            raise NotImplementedError("yolov5_amd.Conv: only SiLU / Identity activations have a fused HIP epilogue")

    def forward(self, x):
        return self._run_single(x, "conv")

    forward_fuse = forward


class Bottleneck(_HipModule):
    """x + cv2(cv1(x)) (models/common.py:164-181)."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return self._run_single(x, "bottleneck")


class C3(_HipModule):
    """CSP bottleneck with 3 convolutions (models/common.py:230-246)."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)))

    def forward(self, x):
        return self._run_single(x, "c3")


class SPPF(_HipModule):
    """Spatial pyramid pooling - fast (models/common.py:318-340)."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        return self._run_single(x, "sppf")


class Concat(nn.Module):
    """Channel concat (models/common.py:443-453).  Inside a model the engine makes it free (producers write into
    slices); the standalone forward is only reachable through a full-model plan."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        raise RuntimeError("yolov5_amd.Concat is executed by the model engine (concat-free slice writes); "
                           "call the parent DetectionModel instead")


class Proto(_HipModule):
    """Mask prototype head (models/common.py:1104-1117)."""

    def __init__(self, c1, c_=256, c2=32):
        super().__init__()
        self.cv1 = Conv(c1, c_, k=3)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        self.cv2 = Conv(c_, c_, k=3)
        self.cv3 = Conv(c_, c2)

    def forward(self, x):
        return self._run_single(x, "proto")


# ----------------------------------------------------------------------------------------------------------
class DetectMultiBackend(nn.Module):
    """Backend selector of the reference (models/common.py:456-814) reduced to its PyTorch (`pt`) branch
    (:485-490, :693-694), which is the drop-in boundary: `weights` may be a DetectionModel instance, a model
    config name ('yolov5s.yaml' -> random init) or a reference *.pt checkpoint."""

    def __init__(self, weights="yolov5s.pt", device=torch.device("cuda"), dnn=False, data=None, fp16=False, fuse=True):
        super().__init__()
        from .yolo import BaseModel, DetectionModel, SegmentationModel
        from .experimental import attempt_load

        w = weights[0] if isinstance(weights, (list, tuple)) else weights
        if isinstance(w, BaseModel):
            model = w
        elif str(w).endswith(".pt"):
            model = attempt_load(w, device=device, fuse=fuse)
        elif str(w).endswith((".yaml",)) or str(w).startswith("yolov5"):
            model = (SegmentationModel if "-seg" in str(w) else DetectionModel)(str(w))
        else:
            raise NotImplementedError(f"DetectMultiBackend: only the PyTorch branch is implemented, got '{w}' "
                                      "(ONNX/TensorRT/TF/... backends are foreign runtimes, out of scope)")
        model = model.to(device).eval()
        if fuse and hasattr(model, "fuse"):
            model.fuse()
        model.half() if fp16 else model.float()
        self.model = model
        self.stride = max(int(model.stride.max()), 32)
        self.names = model.module.names if hasattr(model, "module") else model.names
        self.pt, self.jit, self.onnx, self.engine, self.xml, self.triton = True, False, False, False, False, False
        self.fp16 = fp16
        self.device = torch.device(device)
        self.nhwc = False

    def forward(self, im, augment=False, visualize=False):
        if self.fp16 and im.dtype != torch.float16:
            im = im.half()  # common.py:688-689
        y = self.model(im, augment=augment) if augment else self.model(im)
        return y

    def warmup(self, imgsz=(1, 3, 640, 640)):
        if self.device.type != "cpu":
            im = torch.empty(*imgsz, dtype=torch.half if self.fp16 else torch.float, device=self.device)
            self.forward(im)


class Detections:
    """Tensor part of the reference's results container (models/common.py:950-1101): .xyxy/.xywh/.xyxyn/.xywhn/.pred."""

    def __init__(self, ims, pred, files, times=(0, 0, 0), names=None, shape=None):
        from .general import xyxy2xywh

        d = pred[0].device
        gn = [torch.tensor([*(im.shape[i] for i in [1, 0, 1, 0]), 1, 1], device=d) for im in ims]
        self.ims, self.pred, self.names, self.files, self.times = ims, pred, names, files, times
        self.xyxy = pred
        self.xywh = [xyxy2xywh(x) for x in pred]
        self.xyxyn = [x / g for x, g in zip(self.xyxy, gn)]
        self.xywhn = [x / g for x, g in zip(self.xywh, gn)]
        self.n = len(self.pred)
        self.t = tuple(x / max(self.n, 1) * 1e3 for x in times)
        self.s = tuple(shape) if shape is not None else None

    def tolist(self):
        r = range(self.n)
        return [Detections([self.ims[i]], [self.pred[i]], [self.files[i]], self.times, self.names, self.s) for i in r]

    def __len__(self):
        return self.n

    def __str__(self):
        s = ""
        for i, (im, pred) in enumerate(zip(self.ims, self.pred)):
            s += f"image {i + 1}/{len(self.pred)}: {im.shape[0]}x{im.shape[1]} "
            if pred.shape[0]:
                for c in pred[:, -1].unique():
                    n = int((pred[:, -1] == c).sum())
                    s += f"{n} {self.names[int(c)]}{'s' * (n > 1)}, "
                s = s.rstrip(", ")
            else:
                s += "(no detections)"
            s += "\n"
        return s.rstrip()

    def print(self):
        print(self.__str__())


class AutoShape(nn.Module):
    """Input-robust wrapper (models/common.py:843-948): numpy/PIL/tensor -> letterbox -> model -> NMS -> scale_boxes."""

    conf = 0.25
    iou = 0.45
    agnostic = False
    multi_label = False
    classes = None
    max_det = 1000
    amp = False

    def __init__(self, model, verbose=True):
        super().__init__()
        self.dmb = isinstance(model, DetectMultiBackend)
        self.pt = not self.dmb or model.pt
        self.model = model.eval()
        inner = model.model if self.dmb else model
        self.names = getattr(inner, "names", None)
        self.stride = getattr(model, "stride", None)
        if self.pt:
            m = inner.model[-1]
            m.inplace = False  # common.py:865
            m.export = True  # do not output loss values (common.py:866)

    @torch.no_grad()
    def forward(self, ims, size=640, augment=False, profile=False):
        from .augmentations import letterbox_batch
        from .general import make_divisible, non_max_suppression, scale_boxes_batch

        if isinstance(size, int):
            size = (size, size)
        p = next(self.model.parameters())
        if isinstance(ims, torch.Tensor):  # common.py:900-902 tensor fast path
            return self.model(ims.to(p.device).type_as(p), augment=augment) if self.dmb else self.model(ims.to(p.device).type_as(p))
        n, ims = (len(ims), list(ims)) if isinstance(ims, (list, tuple)) else (1, [ims])
        shape0, shape1, files = [], [], []
        for i, im in enumerate(ims):
            f = f"image{i}"
            if not isinstance(im, np.ndarray):  # PIL
                im, f = np.asarray(im), getattr(im, "filename", f) or f
            files.append(str(f))
            if im.shape[0] < 5:  # CHW -> HWC
                im = im.transpose((1, 2, 0))
            im = im[..., :3] if im.ndim == 3 else np.repeat(im[..., None], 3, 2)
            s = im.shape[:2]
            shape0.append(s)
            g = max(size) / max(s)
            shape1.append([int(y * g) for y in s])
            ims[i] = im if im.data.contiguous else np.ascontiguousarray(im)
        stride = int(max(self.stride)) if hasattr(self.stride, "__len__") or torch.is_tensor(self.stride) else int(self.stride)
        shape1 = [make_divisible(x, stride) for x in np.array(shape1).max(0)]
        # common.py:922-926 (letterbox each image, stack, BHWC -> BCHW, to device, /255) as ONE launch over the raw uint8
        # images: only the un-padded source pixels cross PCIe (augmentations.letterbox_batch -> y5_letterbox_batch)
        raw = [torch.from_numpy(np.array(im, dtype=np.uint8, order="C")).to(p.device) for im in ims]
        x, _ = letterbox_batch(raw, tuple(shape1), auto=False, dtype=p.dtype if p.dtype in (torch.float16, torch.float32) else torch.float32)
        y = self.model(x)
        out, cnt = non_max_suppression(y if self.dmb else y[0], self.conf, self.iou, self.classes, self.agnostic, self.multi_label,
                                       max_det=self.max_det, padded=True)
        scale_boxes_batch(shape1, out, cnt, shape0)  # common.py:940-941 for all images in one launch
        counts = cnt.tolist()  # the one device->host sync of the call
        y = [out[i, :counts[i]] for i in range(n)]
        return Detections(ims, y, files, (0, 0, 0), self.names, x.shape)
