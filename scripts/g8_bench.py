#!/usr/bin/env python
"""Round 6: the 256-row / 8-phase implicit GEMM (conv_g8.h, ids 95..) against the configurations the tuner had, on the MFMA-bound layer shapes of yolov5s
(bs 64, 640^2) and yolov5x (bs 16, 1280^2): N(0,1) activations (cdna_hip_programming.md section 5.4 rule 25: never zero-filled), He-scaled filters,
interleaved A/B rounds in ONE process (rule 24), each arm checked against torch fp32 on the fp16 operands before it is timed."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

# name, B, H(in), C1, C2, k, s, residual
LAYERS = [
    ("5.Conv 3x3s2 128->256 @80", 64, 80, 128, 256, 3, 2, False),
    ("7.Conv 3x3s2 256->512 @40", 64, 40, 256, 512, 3, 2, False),
    ("21.Conv 3x3s2 256->256 @40", 64, 40, 256, 256, 3, 2, False),
    ("9.SPPF.cv2 1x1 1024->512 @20", 64, 20, 1024, 512, 1, 1, False),
    ("8.b.cv2 3x3 256->256 @20 +res", 64, 20, 256, 256, 3, 1, True),
    ("8.cv3 1x1 512->512 @20", 64, 20, 512, 512, 1, 1, False),
    ("6.cv1+cv2 1x1 256->256 @40", 64, 40, 256, 256, 1, 1, False),
    ("18.Conv 3x3s2 128->128 @80", 64, 80, 128, 128, 3, 2, False),
    ("3.Conv 3x3s2 64->128 @160", 64, 160, 64, 128, 3, 2, False),
    ("6.b.cv2 3x3 128->128 @40 +res", 64, 40, 128, 128, 3, 1, True),
    ("x:3x3 80->80 @320 +res", 16, 320, 80, 80, 3, 1, True),
    ("x:3x3s2 80->160 @640", 16, 640, 80, 160, 3, 2, False),
    ("x:3x3 160->160 @160 +res", 16, 160, 160, 160, 3, 1, True),
    ("x:3x3s2 160->320 @320", 16, 320, 160, 320, 3, 2, False),
    ("x:1x1 160->160 @320", 16, 320, 160, 160, 1, 1, False),
    ("x:3x3 320->320 @80 +res", 16, 80, 320, 320, 3, 1, True),
    ("x:3x3s2 320->640 @80", 16, 80, 320, 640, 3, 2, False),
    ("x:3x3 640->640 @40 +res", 16, 40, 640, 640, 3, 1, True),
    ("x:3x3s2 640->1280 @40", 16, 40, 640, 1280, 3, 2, False),
    ("x:1x1 1280->1280 @20", 16, 20, 1280, 1280, 1, 1, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--cfgs", default="95,96", help="the new ids to race")
    ap.add_argument("--base", default="8,12,37,38,39,40,41,42,43,44,46,50,52,64,69,70,73,74,76,77,93,94", help="ids of the round-5 library to race against")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    new = [int(c) for c in a.cfgs.split(",") if c]
    base = [int(c) for c in a.base.split(",") if c]
    rows = []
    for name, B, H, C1, C2, k, s, resid in LAYERS:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        p = k // 2
        OH = (H + 2 * p - k) // s + 1
        torch.manual_seed(0)
        x = torch.randn((B, H, H, C1), device=dev).half()
        w = (torch.randn((C2, C1, k, k), device=dev) * (2.0 / (C1 * k * k)) ** 0.5).half().float()
        b = torch.randn(C2, device=dev) * 0.3
        wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
        r0 = torch.randn((B, OH, OH, C2), device=dev).half()
        ref = F.silu(F.conv2d(x.permute(0, 3, 1, 2).float(), w, b, s, p)).permute(0, 2, 3, 1)
        if resid:
            ref = ref + r0.float()
        flop = 2.0 * B * OH * OH * C2 * C1 * k * k
        y = torch.empty((B, OH, OH, C2), device=dev, dtype=torch.float16)

        def desc(cfg):
            return _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                                 Kpad=Kpad, Npad=Npad, ldr=C2, ld2=0, cfg=cfg, max_blocks=0)

        def run(cfg, iters):
            d = desc(cfg)
            ms = C.c_float(0)
            if resid:
                y.copy_(r0)
            rc = lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()),
                                    C.c_void_p(r0.data_ptr()) if resid else None, C.c_void_p(y.data_ptr()), None, iters, st, C.byref(ms))
            return rc, ms.value

        ok = {}
        for cfg in new + base:
            y.fill_(-3.0)
            d = desc(cfg)
            rc = lib.y5_conv2d_fwd(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()),
                                   C.c_void_p(r0.data_ptr()) if resid else None, C.c_void_p(y.data_ptr()), None, st)
            torch.cuda.synchronize()
            if rc != 0:
                continue
            err = (y.float() - ref).abs().max().item()
            ok[cfg] = err
            if cfg in new and not err < 3e-2:
                print(f"!! {name} cfg {cfg}: max abs err {err:.3e}", flush=True)
        # coarse pass over the base ids, keep the two fastest; then interleaved rounds with the new ids
        coarse = sorted((run(c, 5)[1], c) for c in base if c in ok)
        arms = [c for c in new if c in ok] + [c for _, c in coarse[:2]]
        times = {c: [] for c in arms}
        for _ in range(a.rounds):
            for c in arms:
                times[c].append(run(c, a.iters)[1])
        row = {"layer": name, "gflop": flop / 1e9, "err": {str(c): ok[c] for c in arms},
               "us_median": {str(c): sorted(times[c])[len(times[c]) // 2] * 1e3 for c in arms}, "us_min": {str(c): min(times[c]) * 1e3 for c in arms}}
        row["tflops"] = {c: flop / (v * 1e-6) / 1e12 for c, v in row["us_median"].items()}
        rows.append(row)
        print(f"{name:34s} " + "  ".join(f"[{c}] {row['us_median'][str(c)]:7.1f} us {row['tflops'][str(c)]:6.0f} TF (err {ok[c]:.1e})" for c in arms), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
