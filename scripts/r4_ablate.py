#!/usr/bin/env python
"""Round-4 ablation aid (GPU): what the activation, the depth of K and the grid size cost on the layers that dominate the forward.
For each (layer, cfg): time with act=1 / act=0, with K halved (C1/2), and at max_blocks = 256 / default."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

# name, H, C1, C2, k, s, cfgs
CASES = [
    ("6.cv1+cv2 1x1 256->256 @40", 40, 256, 256, 1, 1, (43, 39)),
    ("17.cv1+cv2 1x1 256->128 @80", 80, 256, 128, 1, 1, (43, 84)),
    ("4.cv3 1x1 128->128 @80", 80, 128, 128, 1, 1, (84, 43)),
    ("6.b.cv1 1x1 128->128 @40", 40, 128, 128, 1, 1, (84, 43)),
    ("3.Conv 3x3s2 64->128 @160", 160, 64, 128, 3, 2, (43,)),
    ("5.Conv 3x3s2 128->256 @80", 80, 128, 256, 3, 2, (43, 39)),
    ("7.Conv 3x3s2 256->512 @40", 40, 256, 512, 3, 2, (39, 43)),
    ("6.b.cv2 3x3 128->128 @40", 40, 128, 128, 3, 1, (76, 70, 43)),
    ("8.b.cv2 3x3 256->256 @20", 20, 256, 256, 3, 1, (76, 39)),
    ("4.b.cv2 3x3 64->64 @80", 80, 64, 64, 3, 1, (80, 76)),
]
lib = _lib.lib()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def run(H, C1, C2, k, s, cfg, act, mb=0, iters=20):
    B, p = 64, k // 2
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn((B, H, H, C1), device=dev, dtype=torch.float16)
    w = torch.randn((C2, C1, k, k), device=dev) * 0.05
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
    y = torch.zeros((B, OH, OH, C2), device=dev, dtype=torch.float16)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=act,
                      Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=mb)
    ms = C.c_float(0)
    rc = lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), None, C.c_void_p(y.data_ptr()), None,
                            iters, st, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")


for name, H, C1, C2, k, s, cfgs in CASES:
    for cfg in cfgs:
        a1 = run(H, C1, C2, k, s, cfg, 1)
        a0 = run(H, C1, C2, k, s, cfg, 0)
        kh = run(H, C1 // 2, C2, k, s, cfg, 1)
        kq = run(H, C1 // 4, C2, k, s, cfg, 1) if C1 >= 128 else float("nan")
        m256 = run(H, C1, C2, k, s, cfg, 1, mb=256)
        print(f"{name:32s} cfg {cfg:2d}: act1 {a1:6.1f} us  act0 {a0:6.1f}  K/2 {kh:6.1f}  K/4 {kq:6.1f}  grid256 {m256:6.1f}", flush=True)
