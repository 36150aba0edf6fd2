"""SPPF pool forward / backward timing at the yolov5s bs=64 shape (20x20, 256 channels in a 1024-channel buffer)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
B, H, W, Cc = 64, 20, 20, 256
buf = torch.randn((B, H, W, 4 * Cc), device=dev).half(); grad = torch.randn_like(buf)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("fwd us", t(lambda: lib.y5_sppf_pool(C.c_void_p(buf.data_ptr()), _lib.Y5_F16, B, H, W, Cc, 4 * Cc, 5, st)))
print("bwd us", t(lambda: lib.y5_sppf_pool_bwd(C.c_void_p(buf.data_ptr()), C.c_void_p(grad.data_ptr()), B, H, W, Cc, 4 * Cc, 4 * Cc, 5, st)))
