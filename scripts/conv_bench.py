#!/usr/bin/env python
"""Per-layer conv micro-benchmark on one MI355X: every tile configuration (y5_conv_cfg_info) on the yolov5s bs=64 640^2
layer shapes, HIP-event timed through y5_conv2d_time.  Prints ms, TFLOP/s and effective GB/s (in + out + filter bytes)."""
import argparse
import ctypes as C
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

# name, H(in), C1, C2, k, s, ldx (channels of the buffer x lives in), ldy, residual
LAYERS = [
    ("1.Conv 3x3s2 32->64 @320", 320, 32, 64, 3, 2, 32, 64, False),
    ("2.cv1+cv2 1x1 64->64 @160", 160, 64, 64, 1, 1, 64, 64, False),
    ("2.b.cv1 1x1 32->32 @160", 160, 32, 32, 1, 1, 64, 32, False),
    ("2.b.cv2 3x3 32->32 @160 +res", 160, 32, 32, 3, 1, 32, 64, True),
    ("2.cv3 1x1 64->64 @160", 160, 64, 64, 1, 1, 64, 64, False),
    ("3.Conv 3x3s2 64->128 @160", 160, 64, 128, 3, 2, 64, 128, False),
    ("4.cv1+cv2 1x1 128->128 @80", 80, 128, 128, 1, 1, 128, 128, False),
    ("4.b.cv1 1x1 64->64 @80", 80, 64, 64, 1, 1, 128, 64, False),
    ("4.b.cv2 3x3 64->64 @80 +res", 80, 64, 64, 3, 1, 64, 128, True),
    ("4.cv3 1x1 128->128 @80", 80, 128, 128, 1, 1, 128, 256, False),
    ("5.Conv 3x3s2 128->256 @80", 80, 128, 256, 3, 2, 256, 256, False),
    ("6.b.cv2 3x3 128->128 @40 +res", 40, 128, 128, 3, 1, 128, 256, True),
    ("6.cv1+cv2 1x1 256->256 @40", 40, 256, 256, 1, 1, 256, 256, False),
    ("17.cv1+cv2 1x1 256->128 @80", 80, 256, 128, 1, 1, 256, 128, False),
    ("14.Conv 1x1 256->128 @40", 40, 256, 128, 1, 1, 256, 256, False),
    ("7.Conv 3x3s2 256->512 @40", 40, 256, 512, 3, 2, 512, 512, False),
    ("9.SPPF.cv2 1x1 1024->512 @20", 20, 1024, 512, 1, 1, 1024, 512, False),
    ("detect.m0 1x1 128->255 @80", 80, 128, 256, 1, 1, 128, 256, False),
    ("8.b.cv2 3x3 256->256 @20 +res", 20, 256, 256, 3, 1, 256, 512, True),
    ("21.Conv 3x3s2 256->256 @40", 40, 256, 256, 3, 2, 256, 512, False),
    ("18.Conv 3x3s2 128->128 @80", 80, 128, 128, 3, 2, 128, 256, False),
    ("8.cv3 1x1 512->512 @20", 20, 512, 512, 1, 1, 512, 512, False),
    ("8.cv1+cv2 1x1 512->512 @20", 20, 512, 512, 1, 1, 512, 512, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ncfg = lib.y5_conv_num_cfgs()
    res = []
    for name, H, C1, C2, k, s, ldx, ldy, resid in LAYERS:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        B = a.batch
        p = k // 2
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn((B, H, H, ldx), device=dev, dtype=torch.float16)
        w = torch.randn((C2, C1, k, k), device=dev) * 0.05
        wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
        y = torch.zeros((B, OH, OH, ldy), device=dev, dtype=torch.float16)
        flop = 2.0 * B * OH * OH * C2 * C1 * k * k
        byts = 2.0 * (B * H * H * C1 + B * OH * OH * C2 * (2 if resid else 1) + C2 * C1 * k * k)
        row = {"layer": name, "gflop": flop / 1e9, "mbytes": byts / 1e6, "cfgs": {}}
        ms = C.c_float(0)
        bm, bn, kb = C.c_int(0), C.c_int(0), C.c_int(0)
        for cfg in range(ncfg):
            lib.y5_conv_cfg_info(cfg, C.byref(bm), C.byref(bn), C.byref(kb))
            if bn.value >= 2 * Npad and bn.value > 32:
                continue
            d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=ldx, OH=OH, OW=OH, C2=C2, ldy=ldy, KH=k, KW=k, SH=s, SW=s,
                              PH=p, PW=p, act=1, Kpad=Kpad, Npad=Npad, ldr=ldy, ld2=0, cfg=cfg, max_blocks=0)
            rc = lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()),
                                    C.c_void_p(y.data_ptr()) if resid else None, C.c_void_p(y.data_ptr()), None, a.iters, st, C.byref(ms))
            if rc == 0:
                row["cfgs"][cfg] = ms.value
        best = min(row["cfgs"], key=row["cfgs"].get)
        t = row["cfgs"][best]
        print(f"{name:34s} best cfg {best:2d} {t:.4f} ms  {flop / t / 1e9:7.1f} TF  {byts / t / 1e6:7.0f} GB/s | " +
              " ".join(f"{c}:{v:.3f}" for c, v in sorted(row["cfgs"].items())), flush=True)
        res.append(row)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
