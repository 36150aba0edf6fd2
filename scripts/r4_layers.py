#!/usr/bin/env python
"""Round-4 A/B aid (GPU): isolated time of given (layer, cfg) pairs through y5_conv2d_time -- run once per library (Y5_LIB_PATH)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

CASES = [
    ("3.Conv 3x3s2 64->128 @160", 160, 64, 128, 3, 2, (43,)),
    ("5.Conv 3x3s2 128->256 @80", 80, 128, 256, 3, 2, (43, 40, 41)),
    ("18.Conv 3x3s2 128->128 @80", 80, 128, 128, 3, 2, (43,)),
    ("21.Conv 3x3s2 256->256 @40", 40, 256, 256, 3, 2, (40, 43)),
    ("7.Conv 3x3s2 256->512 @40", 40, 256, 512, 3, 2, (43, 44, 40)),
    ("6.cv1+cv2 1x1 256->256 @40", 40, 256, 256, 1, 1, (43, 42)),
    ("13.cv1+cv2 1x1 512->256 @40", 40, 512, 256, 1, 1, (43,)),
    ("17.cv1+cv2 1x1 256->128 @80", 80, 256, 128, 1, 1, (43,)),
    ("8.b.cv1 1x1 256->256 @20", 20, 256, 256, 1, 1, (43,)),
    ("9.cv1 1x1 512->256 @20", 20, 512, 256, 1, 1, (43,)),
    ("m1 1x1 256->255 @40", 40, 256, 256, 1, 1, (43,)),
    ("6.b.cv2 3x3 128->128 @40", 40, 128, 128, 3, 1, (43, 76)),
]
lib = _lib.lib()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
tot = 0.0
for name, H, C1, C2, k, s, cfgs in CASES:
    B, p = 64, k // 2
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn((B, H, H, C1), device=dev, dtype=torch.float16)
    w = torch.randn((C2, C1, k, k), device=dev) * 0.05
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
    y = torch.zeros((B, OH, OH, C2), device=dev, dtype=torch.float16)
    out = []
    for cfg in cfgs:
        d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                          Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=0)
        ms = C.c_float(0)
        best = 1e9
        for _ in range(3):
            rc = lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), None, C.c_void_p(y.data_ptr()), None,
                                    20, st, C.byref(ms))
            if rc == 0:
                best = min(best, ms.value * 1e3)
        out.append((cfg, best))
    tot += out[0][1]
    print(f"{name:32s} " + "  ".join(f"cfg {c}: {t:6.1f} us" for c, t in out), flush=True)
print(f"sum of first-listed configurations: {tot:.1f} us")
