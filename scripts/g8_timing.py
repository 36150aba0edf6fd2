#!/usr/bin/env python
"""Kernel-experiment aid (GPU): where the time of a conv_g8.h launch goes, from a library built with -DY5_G8_TIMING (scripts/build_g8_dbg.sh;
Y5_LIB_PATH=yolov5_amd/libyolov5_hip_g8dbg*.so).  Per workgroup and wave row, first output tile: entry -> bias staged -> K tile 0 landed -> K loop done ->
epilogue done (s_memrealtime, 100 MHz), and the shader-clock length of the K loop (s_memtime) -- cycles per K tile against the 2048 cycles its 64 MFMAs
(two waves x 32 x mfma_f32_32x32x16_f16 at 32 cycles) occupy a SIMD's matrix pipe."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

lib = _lib.lib()
lib.y5_g8_dbg_read.restype = C.c_int
lib.y5_g8_dbg_read.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
st = _lib.stream(dev)
SHAPES = {"7": (64, 40, 256, 512, 3, 2), "5": (64, 80, 128, 256, 3, 2), "sppf": (64, 20, 1024, 512, 1, 1), "21": (64, 40, 256, 256, 3, 2), "x640": (16, 40, 640, 640, 3, 1),
          "3": (64, 160, 64, 128, 3, 2), "18": (64, 80, 128, 128, 3, 2), "6cv3": (64, 40, 256, 256, 1, 1)}
for key in (sys.argv[1:] or ["7", "5"]):
    B, H, C1, C2, k, s = SHAPES[key]
    p = k // 2
    OH = (H + 2 * p - k) // s + 1
    torch.manual_seed(0)
    x = torch.randn((B, H, H, C1), device=dev).half()
    w = torch.randn((C2, C1, k, k), device=dev) * (2.0 / (C1 * k * k)) ** 0.5
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
    y = torch.empty((B, OH, OH, C2), device=dev, dtype=torch.float16)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                      Kpad=Kpad, Npad=Npad, ldr=C2, ld2=0, cfg=95, max_blocks=0)
    ms = C.c_float(0)
    vp = lambda t: C.c_void_p(t.data_ptr())
    ts = []
    for _ in range(5):
        _lib.check(lib.y5_conv2d_time(C.byref(d), vp(x), vp(wp), vp(bp), None, vp(y), None, 20, st, C.byref(ms)), lib)
        ts.append(ms.value * 1e3)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (512 * 2 * 8))()
    assert lib.y5_g8_dbg_read(buf) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(512, 2, 8)
    a = a[a[:, 0, 7] > 0]
    nk = int(a[0, 0, 6])
    flop = 2.0 * B * OH * OH * C2 * C1 * k * k
    print(f"shape {key}: {sorted(ts)[2]:.1f} us per launch ({flop / sorted(ts)[2] / 1e6:.0f} TFLOP/s), {len(a)} workgroups, {nk} K tiles per output tile, tiles per workgroup {a[:,0,7].min()}..{a[:,0,7].max()}")
    t0 = a[:, :, 0].min()
    for wr in range(2):
        r = a[:, wr, :]
        seg = np.diff(r[:, :5], axis=1) / 100.0
        cyc = r[:, 5] / nk
        print(f"  wave row {wr}: entry +{(r[:,0].mean() - t0) / 100:.2f} us; bias {seg[:,0].mean():.2f}; K tile 0 landed {seg[:,1].mean():.2f} (max {seg[:,1].max():.2f}); K loop {seg[:,2].mean():.2f} "
              f"(min {seg[:,2].min():.2f} max {seg[:,2].max():.2f}); epilogue {seg[:,3].mean():.2f} (max {seg[:,3].max():.2f}); loop cycles per K tile {cyc.mean():.0f} (min {cyc.min():.0f} max {cyc.max():.0f}) "
              f"= {2048.0 / cyc.mean():.2f} of the matrix pipe; clock {r[:,5].mean() / (seg[:,2].mean() * 1e3):.2f} GHz")
