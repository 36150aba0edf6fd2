#!/usr/bin/env python
"""Kernel-experiment aid (GPU): phase timing inside the streaming 3x3 kernels (conv_k3.h) from a library built with -DY5_K3_TIMING
(Y5_LIB_PATH=yolov5_amd/libyolov5_hip_k3dbg.so): per wave of workgroup 0, shader cycles per tile spent waiting for the stage, issuing the
residual loads, in the MFMA loop, in the epilogue + stores and issuing the refill."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

CASES = [("4.b.cv2 3x3 64->64 @80 +res", 80, 64, 64, 1, True, (32, 78, 79)), ("2.b.cv2 3x3 32->32 @160 +res", 160, 32, 32, 1, True, (30, 33)),
         ("1.Conv 3x3s2 32->64 @320", 320, 32, 64, 2, False, (31, 34))]
lib = _lib.lib()
lib.y5_k3_dbg_read.restype = C.c_int
lib.y5_k3_dbg_read.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for name, H, C1, C2, s, res, cfgs in CASES:
    B, k, p = 64, 3, 1
    OH = (H + 2 - 3) // s + 1
    x = torch.randn((B, H, H, C1), device=dev, dtype=torch.float16)
    w = torch.randn((C2, C1, k, k), device=dev) * 0.05
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
    y = torch.zeros((B, OH, OH, C2), device=dev, dtype=torch.float16)
    for cfg in cfgs:
        d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                          Kpad=Kpad, Npad=Npad, ldr=C2, ld2=0, cfg=cfg, max_blocks=0)
        ms = C.c_float(0)
        yp = C.c_void_p(y.data_ptr())
        rc = lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), yp if res else None, yp, None, 5, st, C.byref(ms))
        if rc:
            print(f"{name} cfg {cfg}: not applicable")
            continue
        torch.cuda.synchronize()
        dbg = (C.c_ulonglong * 64)()
        assert lib.y5_k3_dbg_read(dbg) == 0
        print(f"{name} cfg {cfg}: {ms.value * 1e3:.1f} us")
        for wv in range(4):
            o = dbg[wv * 8: wv * 8 + 6]
            n = max(o[5], 1)
            print(f"   wave {wv}: tiles {o[5]}  cycles per tile: stage wait {o[0] / n:.0f}  residual issue {o[1] / n:.0f}  MFMA loop {o[2] / n:.0f}  epilogue+stores {o[3] / n:.0f}  refill issue {o[4] / n:.0f}")
