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


def k3_ab(a, dev, lib, st):
    tot = {}  # per family
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for H, C1, C2, k, s, cnt in LAYERS:
        if cnt == 0 or k != 3:
            continue
        if a.only and (str(H) not in a.only.split(",") or s != 1):
            continue
        B, OH = a.batch, (H + 2 - 3) // s + 1
        K = 9 * C1
        Kpad, Npad = round_up(K, 64), round_up(C2, 32)
        per = (B * H * H * C1 + B * OH * OH * C2) * 2
        nrot = max(2, -(-600_000_000 // per))
        xs = [torch.randn((B, H, H, C1), device=dev).half() for _ in range(nrot)]
        dzs = [torch.randn((B, OH, OH, C2), device=dev).half() for _ in range(nrot)]
        dw = torch.zeros((Npad, Kpad), device=dev)
        line = f"H={H:3d} {C1:4d}->{C2:4d} k3 s{s} x{cnt}: floor {per / 5.8e12 * 1e6:6.1f} us (HBM) "
        outs = {}
        cfgs = [int(v) for v in a.cfgs.split(",")]
        for cfg in cfgs:
            if cfg == 1:
                tiles = -(-C2 // (128 if C2 >= 128 else 64)) * -(-K // (128 if K >= 128 else 64))
                fs = (1.5, 2, 3, 4, 6)
            else:
                nt, ct = (4 if C2 > 64 else 2 if C2 > 32 else 1), (2 if C1 > 32 else 1)
                if cfg >= 300:   # 3xy: channel tile capped at 32 x by 32 y
                    nt, ct = min(nt, (cfg - 300) // 10), min(ct, (cfg - 300) % 10)
                tiles = -(-C2 // (32 * nt)) * -(-C1 // (32 * ct))
                fs = (0.75, 1, 1.5, 2, 3, 4)
            best = (float("inf"), 0)
            for f in fs:
                mb = max(1, int(f * 256 + tiles - 1) // tiles)
                d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=3, KW=3, SH=s, SW=s, PH=1, PW=1,
                                  act=0, Kpad=Kpad, Npad=Npad, cfg=cfg, max_blocks=mb)
                dw.zero_()
                if a.det:
                    need = lib.y5_conv2d_wgrad_ws_bytes(C.byref(d), C2)
                    ws = torch.empty((need,), dtype=torch.uint8, device=dev)

                    def call(i):
                        return lib.y5_conv2d_wgrad_det(C.byref(d), C.c_void_p(xs[i].data_ptr()), C.c_void_p(dzs[i].data_ptr()), C2, C.c_void_p(dw.data_ptr()),
                                                       C.c_void_p(ws.data_ptr()), need, st)
                else:
                    def call(i):
                        return lib.y5_conv2d_wgrad(C.byref(d), C.c_void_p(xs[i].data_ptr()), C.c_void_p(dzs[i].data_ptr()), C2, C.c_void_p(dw.data_ptr()), st)
                _lib.check(call(0), lib)
                if f == fs[0]:
                    outs[cfg] = dw.clone()
                n = max(a.iters, nrot)
                e0.record()
                for i in range(n):
                    call(i % nrot)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                best = min(best, (us, mb))
            tot[cfg] = tot.get(cfg, 0.0) + best[0] * cnt
            line += f" | cfg{cfg}: {best[0]:7.1f} us (splits {best[1]:4d})"
        if len(cfgs) == 2 and 1 in outs:
            other = outs[[c for c in cfgs if c != 1][0]]
            err = (outs[1] - other).abs().max().item() / max(outs[1].abs().max().item(), 1e-9)
            line += f" | rel diff {err:.1e}"
        print(line, flush=True)
        del xs, dzs
    print("TOTAL 3x3 layers per step: " + ", ".join(f"cfg {c}: {v / 1e3:.3f} ms" for c, v in sorted(tot.items())))


def stem_ab(a, dev, lib, st):
    B, H, W, C2 = a.batch, 640, 320, 32
    OH, OW = 320, 320
    per = (B * H * W * 8 + B * OH * OW * C2) * 2
    nrot = 2
    xs = [torch.randn((B, H, W, 8), device=dev).half() for _ in range(nrot)]
    dzs = [torch.randn((B, OH, OW, C2), device=dev).half() for _ in range(nrot)]
    dw = torch.zeros((32, 192), device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    outs = {}
    for cfg, fs in ((1, (1.5, 2, 3, 4, 6)), (6, (1, 2, 3, 4, 6, 8))):
        best = (float("inf"), 0)
        for f in fs:
            mb = int(f * 256) // (2 if cfg == 1 else 1)
            d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=8, ldx=8, OH=OH, OW=OW, C2=C2, ldy=C2, KH=6, KW=3, SH=2, SW=1, PH=2, PW=1, act=0,
                              Kpad=192, Npad=32, cfg=cfg, max_blocks=mb)
            dw.zero_()
            _lib.check(lib.y5_conv2d_wgrad(C.byref(d), C.c_void_p(xs[0].data_ptr()), C.c_void_p(dzs[0].data_ptr()), C2, C.c_void_p(dw.data_ptr()), st), lib)
            if f == fs[0]:
                outs[cfg] = dw.clone()
            e0.record()
            for i in range(a.iters):
                lib.y5_conv2d_wgrad(C.byref(d), C.c_void_p(xs[i % nrot].data_ptr()), C.c_void_p(dzs[i % nrot].data_ptr()), C2, C.c_void_p(dw.data_ptr()), st)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, (e0.elapsed_time(e1) / a.iters * 1e3, mb))
        print(f"stem wgrad cfg {cfg}: {best[0]:7.1f} us (splits {best[1]}), HBM time {per / 5.8e12 * 1e6:.1f} us", flush=True)
    print("rel diff", ((outs[1] - outs[6]).abs().max() / outs[1].abs().max()).item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--k3-ab", action="store_true", help="3x3 layers only: general gather kernel (cfg 1) against the patch-staged family (cfg 3, csrc/wgrad3.h), "
                    "each at its best split count, on ROTATING buffers (> 600 MB per layer: no Infinity-Cache residency between launches)")
    ap.add_argument("--stem", action="store_true", help="0.Conv's weight gradient (paired-pixel view, 64 x 640 x 320 x 8 -> 32 channels): general kernel (cfg 1) vs the stem kernel (cfg 6)")
    ap.add_argument("--det", action="store_true", help="--k3-ab: time y5_conv2d_wgrad_det (per-split slabs + ordered reduction) instead of the atomic form")
    ap.add_argument("--only", default="", help="--k3-ab: comma list of input sizes H to keep (stride-1 layers only when given)")
    ap.add_argument("--cfgs", default="1,3", help="--k3-ab: kernel families to time")
    ap.add_argument("--split-factors", default="", help="comma list f: also time the pixel-range split count f * CUs / tiles (default of the library: 4)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if a.stem:
        return stem_ab(a, dev, lib, st)
    if a.k3_ab:
        return k3_ab(a, dev, lib, st)
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
