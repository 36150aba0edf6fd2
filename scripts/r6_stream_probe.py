#!/usr/bin/env python
"""Round 6 probe: what bounds the stride-2 3x3 layers with few input channels (3.Conv 64 -> 128 at 160^2, 18.Conv 128 -> 128 at 80^2; bs 64)?  Every
configuration of the library lands at the same ~105 us on 3.Conv (3 TB/s of its unique bytes, 590 TFLOP/s), so the bound is not a tile shape.  Two views:
  * time against the number of workgroups (max_blocks): a bandwidth bound saturates early, a latency bound scales with the workgroup count;
  * with --pmc-run: ONE launch per configuration, for `rocprofv3 --pmc` (FETCH_SIZE / WRITE_SIZE / TCC hits and misses per dispatch).
N(0,1) activations, He-scaled filters; every arm checked against torch fp32 before it is timed."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

LAYERS = [
    ("3.Conv 3x3s2 64->128 @160", 64, 160, 64, 128, 3, 2),
    ("18.Conv 3x3s2 128->128 @80", 64, 80, 128, 128, 3, 2),
    ("5.Conv 3x3s2 128->256 @80", 64, 80, 128, 256, 3, 2),
    ("7.Conv 3x3s2 256->512 @40", 64, 40, 256, 512, 3, 2),
    ("21.Conv 3x3s2 256->256 @40", 64, 40, 256, 256, 3, 2),
    ("4.cv3 1x1 128->128 @80", 64, 80, 128, 128, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="96,95,43")
    ap.add_argument("--blocks", default="0,64,128,192")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--pmc-run", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for name, B, H, C1, C2, k, s in LAYERS:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        p = k // 2
        OH = (H + 2 * p - k) // s + 1
        torch.manual_seed(0)
        x = torch.randn((B, H, H, C1), device=dev).half()
        w = (torch.randn((C2, C1, k, k), device=dev) * (2.0 / (C1 * k * k)) ** 0.5).half().float()
        b = torch.randn(C2, device=dev) * 0.3
        wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
        ref = F.silu(F.conv2d(x.permute(0, 3, 1, 2).float(), w, b, s, p)).permute(0, 2, 3, 1)
        y = torch.empty((B, OH, OH, C2), device=dev, dtype=torch.float16)
        flop = 2.0 * B * OH * OH * C2 * C1 * k * k
        unique = x.numel() * 2 + y.numel() * 2 + wp.numel() * 2

        def desc(cfg, mb):
            return _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                                 Kpad=Kpad, Npad=Npad, ldr=C2, ld2=0, cfg=cfg, max_blocks=mb)

        print(f"{name}: {flop / 1e9:.1f} GFLOP, unique bytes {unique / 1e6:.0f} MB (8 TB/s: {unique / 8e6:.1f} us)", flush=True)
        for cfg in [int(c) for c in a.cfgs.split(",")]:
            for mb in [int(v) for v in a.blocks.split(",")]:
                d = desc(cfg, mb)
                y.fill_(-3.0)
                rc = lib.y5_conv2d_fwd(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), None, C.c_void_p(y.data_ptr()), None, st)
                torch.cuda.synchronize()
                if rc != 0:
                    print(f"  cfg {cfg} blocks {mb}: rc {rc}")
                    continue
                err = (y.float() - ref).abs().max().item()
                if a.pmc_run:
                    print(f"  cfg {cfg} blocks {mb}: one launch (err {err:.1e})", flush=True)
                    continue
                ts = []
                for _ in range(3):
                    ms = C.c_float(0)
                    lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), None, C.c_void_p(y.data_ptr()), None,
                                       a.iters, st, C.byref(ms))
                    ts.append(ms.value * 1e3)
                us = sorted(ts)[1]
                print(f"  cfg {cfg:3d} blocks {mb:4d}: {us:7.1f} us  {flop / us / 1e6:6.0f} TF  {unique / us / 1e6:5.2f} TB/s of unique bytes  (err {err:.1e})", flush=True)


if __name__ == "__main__":
    main()
