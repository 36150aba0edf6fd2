#!/usr/bin/env python
"""Kernel-experiment aid (GPU): phase timing inside the fused 128-channel Bottleneck kernel (conv_h3b.h) from a library built with
-DY5_H3B_TIMING (Y5_LIB_PATH=yolov5_amd/libyolov5_hip_h3bdbg.so, scripts/build_h3b_dbg.sh).  Per workgroup and tile: s_memrealtime at
tile start / x halo landed / GEMM 1 done / t stored + first slice landed / tap loop done / epilogue done."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

lib = _lib.lib()
lib.y5_h3b_dbg_read.restype = C.c_int
lib.y5_h3b_dbg_read.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
st = _lib.stream(dev)
vp = lambda t: C.c_void_p(t.data_ptr())
Cc, HW, B = 128, 40, 64
torch.manual_seed(0)
w1 = torch.randn(Cc, Cc, 1, 1) * (2.0 / Cc) ** 0.5; b1 = torch.randn(Cc) * 0.1
w2 = torch.randn(Cc, Cc, 3, 3) * (2.0 / (9 * Cc)) ** 0.5; b2 = torch.randn(Cc) * 0.1
w1p, b1p, _, K1, N1 = pack_conv_weight(w1, b1, torch.float16)
w2p, b2p, _, K2, N2 = pack_conv_weight(w2, b2, torch.float16)
w1p, b1p, w2p, b2p = (t.to(dev) for t in (w1p, b1p, w2p, b2p))
cat = torch.randn(B, HW, HW, 2 * Cc, device=dev).half()
out = torch.empty(B, HW, HW, Cc, device=dev, dtype=torch.float16)
for mb in [int(a, 0) for a in (sys.argv[1:] or ["0", str(18 << 16), str(5 << 16)])]:
    f = lambda: _lib.check(lib.y5_bottleneck_fwd(vp(cat), 2 * Cc, vp(w1p), vp(b1p), K1, vp(w2p), vp(b2p), K2, vp(out), Cc, B, HW, HW, Cc, 1, mb, st), lib)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (512 * 4 * 8))()
    assert lib.y5_h3b_dbg_read(buf) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(512, 4, 8)[:, :, :6]
    print(f"stages {mb >> 16} grid cap {mb & 0xffff}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch")
    blocks = a[(a[:, 0, 5] > 0)]
    t0 = blocks[:, 0, 0].min()
    for ti in range(4):
        tl = blocks[blocks[:, ti, 5] > blocks[:, ti, 0]][:, ti, :]
        if not len(tl) or (ti and tl[:, 0].min() < t0):
            continue
        d = np.diff(tl, axis=1) / 100.0
        names = ["wait x", "GEMM 1", "t store + slice 0", "tap loop", "epilogue"]
        print(f"  tile {ti}: {len(tl)} workgroups, start {(tl[:,0].mean() - t0) / 100:.1f} us, end {(tl[:,5].mean() - t0) / 100:.1f} (max {(tl[:,5].max() - t0) / 100:.1f}); "
              + "; ".join(f"{n} {d[:, k].mean():.2f}" for k, n in enumerate(names)) + " us")
