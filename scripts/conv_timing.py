#!/usr/bin/env python
"""Kernel-experiment aid (GPU): phase timing (s_memtime) inside the 2-stage implicit-GEMM mainloop of workgroup 0, from a
library built with -DY5_DBG_TIMING (Y5_LIB_PATH).  Prints cycles per chunk spent issuing loads, in the epilogue, in the
LDS-read + MFMA block, waiting for vmcnt(0) and at the barrier."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

CASES = [("7.Conv 3x3s2 256->512 @40", 40, 256, 512, 3, 2), ("6.b.cv2 3x3 128->128 @40", 40, 128, 128, 3, 1), ("5.Conv 3x3s2 128->256 @80", 80, 128, 256, 3, 2),
         ("9.SPPF.cv2 1x1 1024->512 @20", 20, 1024, 512, 1, 1)]
lib = _lib.lib()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for name, H, C1, C2, k, s in CASES:
    B, p = 64, k // 2
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn((B, H, H, C1), device=dev, dtype=torch.float16)
    w = torch.randn((C2, C1, k, k), device=dev) * 0.05
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
    y = torch.zeros((B, OH, OH, C2), device=dev, dtype=torch.float16)
    for cfg in (8, 39, 2):
        d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                          Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=0)
        ms = C.c_float(0)
        rc = lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), None, C.c_void_p(y.data_ptr()), None,
                                5, st, C.byref(ms))
        if rc:
            continue
        buf = (C.c_ulonglong * 128)()
        lib.y5_dbg_read_timing(buf)
        torch.cuda.synchronize()
        print(f"{name} cfg {cfg}: {ms.value * 1e3:.1f} us")
        for wv in range(8):
            o = buf[wv * 8: wv * 8 + 8]
            if o[6] == 0:
                continue
            n = o[6]
            print(f"   wave {wv}: chunks {n} tiles {o[7]}  per chunk [100MHz ticks]: stage {o[0] / n:.1f} epi {o[1] / n:.1f} compute {o[2] / n:.1f} vmwait {o[3] / n:.1f} barrier {o[4] / n:.1f}"
                  f"  total {o[5] / n:.1f}  | shader clock {o[5] / max(buf[64 + wv], 1) * 100:.0f} MHz, loop {buf[64 + wv] / 100:.1f} us")
        bb = (C.c_ulonglong * 4096)()
        lib.y5_dbg_read_blocks(bb)
        import numpy as np
        arr = np.array(bb[:], dtype=np.int64).reshape(1024, 4)
        arr = arr[arr[:, 3] > 0]
        if len(arr):
            t0 = arr[:, 0].min()
            a = (arr - t0) / 100.0  # us
            print(f"   blocks {len(arr)}: entry min/max {a[:,0].min():.1f}/{a[:,0].max():.1f} us; loop start mean {a[:,1].mean():.1f}; loop end mean {a[:,2].mean():.1f} "
                  f"min {a[:,2].min():.1f} max {a[:,2].max():.1f}; exit mean {a[:,3].mean():.1f} max {a[:,3].max():.1f}; prologue mean {(a[:,1]-a[:,0]).mean():.2f} final-epilogue mean {(a[:,3]-a[:,2]).mean():.2f}")
        for i in range(128):
            buf[i] = 0
