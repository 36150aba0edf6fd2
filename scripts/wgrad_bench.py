#!/usr/bin/env python
"""Per-layer microbenchmark of y5_conv2d_wgrad on the yolov5s bs=64 640^2 training shapes (GPU).  Y5_LIB_PATH selects the
library build (ablation variants: -DY5_WG_NOSTAGE / NOREAD / NOMFMA / NOATOM)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov5_amd import _lib  # noqa: E402
from yolov5_amd.packing import round_up  # noqa: E402

# (H, C1, C2, k, s, count) of yolov5s at 640^2: input H, channels, kernel, stride, number of such layers
LAYERS = [
    (320, 32, 64, 3, 2, 1), (160, 64, 32, 1, 1, 2), (160, 32, 32, 1, 1, 1), (160, 32, 32, 3, 1, 1), (160, 64, 64, 1, 1, 1),
    (160, 64, 128, 3, 2, 1), (80, 128, 64, 1, 1, 2), (80, 64, 64, 1, 1, 2), (80, 64, 64, 3, 1, 2), (80, 128, 128, 1, 1, 1),
    (80, 128, 256, 3, 2, 1), (40, 256, 128, 1, 1, 2), (40, 128, 128, 1, 1, 3), (40, 128, 128, 3, 1, 3), (40, 256, 256, 1, 1, 1),
    (40, 256, 512, 3, 2, 1), (20, 512, 256, 1, 1, 2), (20, 256, 256, 1, 1, 1), (20, 256, 256, 3, 1, 1), (20, 512, 512, 1, 1, 1),
    (20, 512, 256, 1, 1, 1), (20, 1024, 512, 1, 1, 1), (80, 256, 64, 1, 1, 2), (80, 128, 128, 3, 2, 1), (40, 256, 128, 3, 2, 0),
    (40, 256, 256, 3, 2, 1), (80, 128, 256, 1, 1, 1), (40, 256, 256, 1, 1, 1), (20, 512, 256, 1, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--split-factors", default="", help="comma list f: also time the pixel-range split count f * CUs / tiles (default of the library: 4)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    tot_ms, tot_fl = 0.0, 0.0
    for H, C1, C2, k, s, cnt in LAYERS:
        if cnt == 0:
            continue
        p = k // 2
        OH = (H + 2 * p - k) // s + 1
        B = a.batch
        x = torch.randn((B, H, H, C1), device=dev).half()
        dz = torch.randn((B, OH, OH, C2), device=dev).half()
        K = k * k * C1
        Kpad, Npad = round_up(K, 64), round_up(C2, 32)
        dw = torch.zeros((Npad, Kpad), device=dev)
        d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p,
                          act=0, Kpad=Kpad, Npad=Npad, cfg=-1, max_blocks=0)
        args = (C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(dz.data_ptr()), C2, C.c_void_p(dw.data_ptr()), st)
        _lib.check(lib.y5_conv2d_wgrad(*args), lib)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            lib.y5_conv2d_wgrad(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * B * OH * OH * C2 * K
        gb = (x.numel() + dz.numel()) * 2 / 1e9
        tot_ms += ms * cnt
        tot_fl += fl * cnt
        extra = ""
        for f in [float(v) for v in a.split_factors.split(",") if v]:
            tiles = -(-C2 // (128 if C2 >= 128 else 64)) * -(-K // (128 if K >= 128 else 64))
            d.max_blocks = max(1, int(f * 256 + tiles - 1) // tiles)
            lib.y5_conv2d_wgrad(*args)
            e0.record()
            for _ in range(a.iters):
                lib.y5_conv2d_wgrad(*args)
            e1.record()
            torch.cuda.synchronize()
            extra += f"  f={f:g}:{e0.elapsed_time(e1) / a.iters * 1e3:.1f}"
        d.max_blocks = 0
        print(f"H={H:3d} {C1:4d}->{C2:4d} k{k} s{s} x{cnt}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF  {gb / ms * 1e3:6.2f} TB/s(min traffic){extra}")
    print(f"TOTAL {tot_ms:.3f} ms per step, {tot_fl / tot_ms / 1e9:.1f} TF")


if __name__ == "__main__":
    main()
