"""CPU: the C-ABI library is built, loads, and exports every symbol include/yolov5_hip.h declares; the host-side
argument validation of the reference-facing wrappers behaves like the reference's assertions."""
import ctypes
import os
import re

import pytest
import torch

from yolov5_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_match_binding_table():
    hdr = open(os.path.join(ROOT, "include", "yolov5_hip.h")).read()
    declared = set(re.findall(r"\b(y5_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"y5_status", "y5_dtype", "y5_conv_desc", "y5_plan"}  # (types, not functions)
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_library_loads_and_exports_everything():
    so = _lib.LIB_PATH
    if not os.path.isfile(so):
        import __graft_entry__

        __graft_entry__.build()
    try:
        lib = _lib.bind(ctypes.CDLL(so))
    except OSError as e:
        pytest.skip(f"HIP runtime not loadable on this host: {e}")
    assert lib.y5_version() >= 100
    assert lib.y5_plan_size(None) == 0


def test_nms_argument_checks_match_reference_messages():
    from yolov5_amd.general import non_max_suppression

    p = torch.zeros((1, 10, 85))
    with pytest.raises(AssertionError, match="Invalid Confidence threshold"):
        non_max_suppression(p, conf_thres=1.5)
    with pytest.raises(AssertionError, match="Invalid IoU"):
        non_max_suppression(p, iou_thres=-0.1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        non_max_suppression(p)


def test_no_cpu_fallback_for_model_forward():
    from yolov5_amd.yolo import DetectionModel

    m = DetectionModel("yolov5n.yaml").eval()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="GPU"):
        m.train()(torch.zeros(1, 3, 64, 64))
