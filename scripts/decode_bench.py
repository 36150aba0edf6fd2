"""Detect decode timing at the yolov5s bs=64 shapes (z only, fp16)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
B = 64
z = torch.zeros((B, 25200, 85), dtype=torch.float16, device=dev)
anchors = (C.c_float * 6)(10, 13, 16, 30, 33, 23)
off = 0
for ny, stride in ((80, 8.0), (40, 16.0), (20, 32.0)):
    lg = torch.randn((B, ny, ny, 256), device=dev).half()
    args = (C.c_void_p(lg.data_ptr()), _lib.Y5_F16, B, ny, ny, 3, 85, 0, 256, stride, anchors, C.c_void_p(z.data_ptr()), _lib.Y5_F16, 25200, off, None, st)
    for _ in range(3): _lib.check(lib.y5_detect_decode(*args), lib)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.y5_detect_decode(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (lg.numel() + B * 3 * ny * ny * 85) * 2 / 1e6
    print(f"decode {ny}x{ny}: {us:.1f} us  ({mb / us * 1e-3 * 1e3:.2f} GB/s... {mb:.0f} MB -> {mb / us / 1e3:.2f} TB/s)")
    off += 3 * ny * ny
