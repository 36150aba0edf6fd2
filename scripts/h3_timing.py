#!/usr/bin/env python
"""Kernel-experiment aid (GPU): phase timing inside the halo-resident 3x3 kernel (conv_h3.h) from a library built with
-DY5_H3_TIMING (Y5_LIB_PATH=yolov5_amd/libyolov5_hip_h3dbg.so).  Per wave of workgroup 0: cycles per step spent waiting
(vmcnt + barrier), issuing LDS-DMA, and in the fragment-read + MFMA block; per workgroup: entry / first step / loop end / exit."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

CASES = [("6.b.cv2 3x3 128->128 @40", 40, 128, 128), ("8.b.cv2 3x3 256->256 @20", 20, 256, 256), ("4.b.cv2 3x3 64->64 @80", 80, 64, 64)]
lib = _lib.lib()
lib.y5_h3_dbg_read.restype = C.c_int
lib.y5_h3_dbg_read.argtypes = [C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "61,63,64,67,68,69,70".split(","))]
for name, H, C1, C2 in CASES:
    B, k, s, p = 64, 3, 1, 1
    x = torch.randn((B, H, H, C1), device=dev, dtype=torch.float16)
    w = torch.randn((C2, C1, k, k), device=dev) * 0.05
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
    y = torch.zeros((B, H, H, C2), device=dev, dtype=torch.float16)
    for cfg in cfgs:
        d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=H, OW=H, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                          Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=0)
        ms = C.c_float(0)
        rc = lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), None, C.c_void_p(y.data_ptr()), None,
                                5, st, C.byref(ms))
        if rc:
            print(f"{name} cfg {cfg}: not applicable ({lib.y5_last_error().decode()})")
            continue
        torch.cuda.synchronize()
        dbg = (C.c_ulonglong * 64)()
        blk = (C.c_ulonglong * 4096)()
        assert lib.y5_h3_dbg_read(dbg, blk) == 0
        print(f"{name} cfg {cfg}: {ms.value * 1e3:.1f} us")
        for wv in range(8):
            o = dbg[wv * 8: wv * 8 + 4]
            if o[3] == 0:
                continue
            n = o[3]
            print(f"   wave {wv}: steps {n}  per step [s_memtime ticks]: wait+barrier {o[0] / n:.0f} issue {o[1] / n:.0f} read+mfma {o[2] / n:.0f}")
        arr = np.array(blk[:], dtype=np.int64).reshape(1024, 4)
        arr = arr[arr[:, 3] > 0]
        if len(arr):
            t0 = arr[:, 0].min()
            a = (arr - t0) / 100.0  # us
            print(f"   blocks {len(arr)}: entry min/max {a[:,0].min():.1f}/{a[:,0].max():.1f} us; first step mean {a[:,1].mean():.1f}; loop end mean {a[:,2].mean():.1f} "
                  f"min {a[:,2].min():.1f} max {a[:,2].max():.1f}; exit mean {a[:,3].mean():.1f} max {a[:,3].max():.1f}; prologue mean {(a[:,1]-a[:,0]).mean():.2f} "
                  f"loop mean {(a[:,2]-a[:,1]).mean():.2f} epilogue mean {(a[:,3]-a[:,2]).mean():.2f}")
