"""yolov5_amd -- MI355X (gfx950) native YOLOv5 hot path: forward, non_max_suppression, ComputeLoss.

Host side mirrors the reference's Python operator API (models/yolo.py, models/common.py, utils/general.py,
utils/loss.py) and calls hand-written HIP kernels through the C-ABI in include/yolov5_hip.h.
There is NO CPU fallback: every op raises if libyolov5_hip.so is missing or a tensor is not on a GPU.
"""
__version__ = "0.1.0"
