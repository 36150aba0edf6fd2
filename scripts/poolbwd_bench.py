"""Kernel-experiment helper: y5_sppf_pool_bwd at yolov5s' 9.SPPF training shape (64 x 20 x 20 x 256, k = 5), average of 50 launches (torch events on the
current stream = the stream the launch goes to).  Y5_SPPF_BWD_GV / Y5_SPPF_BWD_GATHER select the form (latched per process)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from yolov5_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
B, H, W, Cc = 64, 20, 20, 256
act = torch.randn((B, H, W, 4 * Cc), device=dev).half()
grad0 = torch.randn((B, H, W, 4 * Cc), device=dev).half()
grad = grad0.clone()
st = _lib.stream(dev)
for _ in range(3):
    _lib.check(lib.y5_sppf_pool_bwd(C.c_void_p(act.data_ptr()), C.c_void_p(grad.data_ptr()), B, H, W, Cc, 4 * Cc, 4 * Cc, 5, st), lib)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    lib.y5_sppf_pool_bwd(C.c_void_p(act.data_ptr()), C.c_void_p(grad.data_ptr()), B, H, W, Cc, 4 * Cc, 4 * Cc, 5, st)
e1.record()
torch.cuda.synchronize()
print(f"GV={os.environ.get('Y5_SPPF_BWD_GV', 'auto')} GATHER={os.environ.get('Y5_SPPF_BWD_GATHER', '0')}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch")
