"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (yolov5_amd/).

Import shim that loads the *unmodified* reference modules from /root/reference
(`models/yolo.py`, `models/common.py`, `utils/loss.py`, `utils/general.py`,
`utils/segment/general.py`) on torch-CPU, in THIS container only.  The reference
imports `cv2`, `torchvision` and the un-vendored `ultralytics>=8.4.118` pip package
(requirements.txt:15-16) at module top; none is installed here.  We inject stub modules
(the same trick the reference's own tests use, tests/test_invariant_export.py:47-71) and
supply restated implementations for the handful of third-party functions the hot path
actually calls.  Those restatements live in `oracle/thirdparty.py`.

`/root/reference` does not exist on the GPU box: only `oracle/make_golden.py` and the
`-m "not gpu"` cross-check tests (skipped when the directory is absent) may use this file.
"""
from __future__ import annotations

import contextlib
import logging
import math
import os
import sys
import threading
import time
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get("YOLOV5_REFERENCE_ROOT", "/root/reference")

_loaded = None


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "yolo.py"))


def _stub(name: str) -> types.ModuleType:
    m = MagicMock(name=name)
    m.__name__ = name
    m.__path__ = []  # behave like a package so that `import a.b.c` works
    m.__spec__ = None
    sys.modules[name] = m
    return m


def load():
    """Import the reference under stubs; returns a namespace with the modules we need."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

    import torch
    from torch import nn

    from . import thirdparty as tp

    for name in [
        "cv2",
        "torchvision", "torchvision.ops", "torchvision.transforms", "torchvision.models",
        "ultralytics",
        "ultralytics.utils", "ultralytics.utils.plotting", "ultralytics.utils.checks",
        "ultralytics.utils.files", "ultralytics.utils.git", "ultralytics.utils.ops",
        "ultralytics.utils.patches", "ultralytics.utils.torch_utils", "ultralytics.utils.metrics",
        "ultralytics.utils.export", "ultralytics.utils.export.tensorflow", "ultralytics.utils.downloads",
        "ultralytics.utils.errors", "ultralytics.utils.loss", "ultralytics.utils.autobatch",
        "ultralytics.data", "ultralytics.data.converter", "ultralytics.data.build",
        "ultralytics.data.utils", "ultralytics.data.loaders", "ultralytics.data.augment",
        "ultralytics.nn", "ultralytics.nn.modules", "ultralytics.nn.autobackend",
        "pandas_stub_unused",
    ]:
        if name not in sys.modules:
            _stub(name)

    sys.modules["ultralytics"].__version__ = "8.4.118"
    sys.modules["cv2"].__version__ = "4.6.0"
    sys.modules["cv2"].setNumThreads = lambda *_a, **_k: None
    sys.modules["cv2"].INTER_LINEAR, sys.modules["cv2"].BORDER_CONSTANT = 1, 0
    sys.modules["cv2"].resize = tp.cv2_resize  # restated, parity unpinned (oracle/thirdparty.py)
    sys.modules["cv2"].copyMakeBorder = tp.cv2_copy_make_border
    cv2m = sys.modules["cv2"]  # training augmentations (restated, parity unpinned): warpAffine, HSV conversion, LUT
    cv2m.INTER_AREA, cv2m.COLOR_BGR2HSV, cv2m.COLOR_HSV2BGR = 3, tp.COLOR_BGR2HSV, tp.COLOR_HSV2BGR
    cv2m.getRotationMatrix2D = lambda angle=0, center=(0, 0), scale=1.0: tp.cv2_get_rotation_matrix_2d(center, angle, scale)
    cv2m.warpAffine = tp.cv2_warp_affine
    cv2m.cvtColor, cv2m.LUT, cv2m.split, cv2m.merge = tp.cv2_cvt_color, tp.cv2_lut, tp.cv2_split, tp.cv2_merge
    sys.modules["torchvision"].__version__ = "0.19.0"

    # --- real implementations for the symbols the hot path touches -------------------
    u = sys.modules["ultralytics.utils"]
    logger = logging.getLogger("yolov5-oracle")
    logger.setLevel(logging.WARNING)
    u.LOGGER = logger
    u.colorstr = lambda *a: str(a[-1])
    u.emojis = lambda s="": s
    u.TQDM = MagicMock()
    u.RANK = -1

    class TryExcept(contextlib.ContextDecorator):
        def __init__(self, msg="", verbose=True):
            self.msg = msg

        def __enter__(self):
            return self

        def __exit__(self, et, ev, tb):
            return True

    u.TryExcept = TryExcept

    class WorkingDirectory(contextlib.ContextDecorator):
        def __init__(self, new_dir):
            self.dir, self.cwd = new_dir, os.getcwd()

        def __enter__(self):
            os.chdir(self.dir)

        def __exit__(self, *a):
            os.chdir(self.cwd)

    sys.modules["ultralytics.utils.files"].WorkingDirectory = WorkingDirectory

    def threaded(func):
        def wrapper(*a, **k):
            t = threading.Thread(target=func, args=a, kwargs=k, daemon=True)
            t.start()
            return t

        return wrapper

    u.threaded = threaded

    ops = sys.modules["ultralytics.utils.ops"]
    ops.xywh2xyxy = tp.xywh2xyxy
    ops.clip_boxes = tp.clip_boxes
    ops.make_divisible = tp.make_divisible

    class Profile(contextlib.ContextDecorator):
        def __init__(self, t=0.0, device=None):
            self.t, self.dt = t, 0.0

        def __enter__(self):
            self.s = time.time()
            return self

        def __exit__(self, *a):
            self.dt = time.time() - self.s
            self.t += self.dt

    ops.Profile = Profile

    met = sys.modules["ultralytics.utils.metrics"]
    met.bbox_iou = tp.bbox_iou
    met.box_iou = tp.box_iou
    met.smooth_bce = tp.smooth_bce
    met.smooth = tp.smooth
    met.bbox_ioa = None  # copy-paste augmentation: not on the path
    met.mask_iou = None  # mask branch of process_batch: not on the path
    met.plot_mc_curve = met.plot_pr_curve = lambda *a, **k: None

    tu = sys.modules["ultralytics.utils.torch_utils"]
    tu.initialize_weights = tp.initialize_weights
    tu.model_info = lambda *a, **k: None
    tu.time_sync = time.time
    tu.scale_img = tp.scale_img
    tu.copy_attr = tp.copy_attr
    tu.autocast = lambda enabled, device="cpu": contextlib.nullcontext()
    tu.is_parallel = lambda m: isinstance(m, (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel))
    tu.de_parallel = lambda m: m.module if tu.is_parallel(m) else m
    tu.get_flops = lambda *a, **k: 0.0
    tu.intersect_dicts = lambda da, db, exclude=(): {
        k: v for k, v in da.items() if k in db and all(x not in k for x in exclude) and v.shape == db[k].shape
    }
    tu.one_cycle = lambda y1=0.0, y2=1.0, steps=100: (
        lambda x: max((1 - math.cos(x * math.pi / steps)) / 2, 0) * (y2 - y1) + y1
    )
    u.checks = sys.modules["ultralytics.utils.checks"]
    sys.modules["ultralytics.utils.checks"].check_version = lambda *a, **k: True
    sys.modules["ultralytics.utils.checks"].check_requirements = lambda *a, **k: True
    sys.modules["ultralytics.utils.patches"].torch_load = torch.load

    # torchvision.ops.nms -> restated greedy NMS
    sys.modules["torchvision.ops"].nms = tp.nms
    sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]

    # yolov5_amd.experimental.install_reference_aliases (checkpoint save / load) registers ALIAS modules `models.yolo` / `models.common` whose
    # classes are yolov5_amd's: a test that ran earlier in this process may have left them -- the reference must import its own files
    if getattr(sys.modules.get("models.yolo"), "__y5amd__", False):
        for name in [n for n in sys.modules if n == "models" or n.startswith("models.")]:
            del sys.modules[name]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)  # reference resolves ROOT relative to cwd (models/yolo.py:24-25)
    try:
        import models.common as ref_common  # noqa
        import models.yolo as ref_yolo  # noqa
        import utils.general as ref_general  # noqa
        import utils.loss as ref_loss  # noqa
        import utils.torch_utils as ref_torch_utils  # noqa
        import utils.segment.general as ref_seg_general  # noqa
        import utils.metrics as ref_metrics  # noqa
        import utils.augmentations as ref_aug  # noqa
    finally:
        os.chdir(cwd)

    # make sure the functions bound at import time are the restated ones
    ref_general.xywh2xyxy = tp.xywh2xyxy
    ref_general.clip_boxes = tp.clip_boxes
    ref_general.torchvision = sys.modules["torchvision"]
    ref_loss.bbox_iou = tp.bbox_iou
    ref_loss.smooth_bce = tp.smooth_bce
    ref_yolo.check_version = lambda *a, **k: True
    ref_yolo.initialize_weights = tp.initialize_weights
    ref_yolo.make_divisible = tp.make_divisible
    ref_yolo.scale_img = tp.scale_img
    ref_yolo.model_info = lambda *a, **k: None

    ns = types.SimpleNamespace(
        common=ref_common,
        yolo=ref_yolo,
        general=ref_general,
        loss=ref_loss,
        torch_utils=ref_torch_utils,
        seg_general=ref_seg_general,
        metrics=ref_metrics,
        augmentations=ref_aug,
        root=REFERENCE_ROOT,
    )
    _loaded = ns
    return ns


def unload():
    """Forget the imported reference: its `models` / `utils` packages leave sys.modules and its root leaves sys.path, so that code which must NOT
    see a live reference checkout in the process (yolov5_amd.experimental.install_reference_aliases refuses to alias `models.yolo` then) behaves
    as on a clean interpreter.  The stub modules (cv2, torchvision, ultralytics.*) stay: they are harmless.  Test modules that load() the reference
    call this when they are done (module-scoped fixture finaliser) -- pytest-xdist may run any other test file in the same worker afterwards."""
    global _loaded
    _loaded = None
    for name in list(sys.modules):
        if name in ("models", "utils", "export", "val", "train", "detect") or name.startswith(("models.", "utils.")):
            mod = sys.modules.get(name)
            f = getattr(mod, "__file__", None) or ""
            if f.startswith(REFERENCE_ROOT) or (not f and name.split(".")[0] in ("models", "utils") and not getattr(mod, "__y5amd__", False)
                                                and not getattr(sys.modules.get("models.yolo"), "__y5amd__", False)):
                del sys.modules[name]
    while REFERENCE_ROOT in sys.path:
        sys.path.remove(REFERENCE_ROOT)


@contextlib.contextmanager
def oracle_nms_mode():
    """Context that removes the two time/implementation dependent behaviours of the reference NMS.

    (1) general.py:692,763-765 wall-clock `time_limit` break -> disabled by freezing time.time inside
        utils.general.  (2) general.py:745 `argsort(descending=True)` is not stable on torch-CPU for large
        inputs -> patched to stable=True (tie rule: lower original row first; SURVEY 8c hazards).
    """
    import torch

    ns = load()
    g = ns.general
    real_time = g.time
    frozen = types.SimpleNamespace(time=lambda: 0.0)
    real_argsort = torch.Tensor.argsort

    def stable_argsort(self, *a, **k):
        k.setdefault("stable", True)
        return real_argsort(self, *a, **k)

    g.time = frozen
    torch.Tensor.argsort = stable_argsort
    try:
        yield ns
    finally:
        g.time = real_time
        torch.Tensor.argsort = real_argsort
