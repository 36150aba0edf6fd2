#!/usr/bin/env python
"""Why are the deep layers 10-25 % slower IN SITU than in 10-launch bursts (op table: ms vs ms_isolated)?  One conv launch timed under four
cache states, each state prepared by ordinary kernels right before the launch:

  warm          back-to-back launches of the same op (what `ms_isolated` measures)
  cold          a 1 GiB device copy first: L2 and the 256 MiB Infinity Cache hold neither the input nor the filter
  input_warm    cold, then the INPUT is read once (x.sum()): the state a layer finds in the forward -- its producer has just written the
                activation, its filter was last used a whole forward (7 GB of traffic) ago
  both_warm     cold, then input and FILTER are read once

input_warm - both_warm = what a filter prefetch (issued while the previous layer runs) would return per launch.
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scripts.conv_bench import LAYERS
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--only", default="7.Conv,21.Conv,8.cv1+cv2,9.SPPF.cv2,8.b.cv2,6.b.cv2,5.Conv,14.Conv,8.cv3")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    big_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    big_b = torch.empty_like(big_a)
    ncfg = lib.y5_conv_num_cfgs()
    out = []
    for name, H, C1, C2, k, s, ldx, ldy, resid in LAYERS:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        B, p = a.batch, k // 2
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn((B, H, H, ldx), device=dev, dtype=torch.float16)
        w = torch.randn((C2, C1, k, k), device=dev) * 0.05
        wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
        y = torch.zeros((B, OH, OH, ldy), device=dev, dtype=torch.float16)
        ptrs = (C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), C.c_void_p(y.data_ptr()) if resid else None,
                C.c_void_p(y.data_ptr()), None)
        mk = lambda cfg: _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=ldx, OH=OH, OW=OH, C2=C2, ldy=ldy, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p,
                                       act=1, Kpad=Kpad, Npad=Npad, ldr=ldy, ld2=0, cfg=cfg, max_blocks=0)  # noqa: E731
        ms = C.c_float(0)
        best, best_ms = -1, 1e9
        bm, bn, kb = C.c_int(0), C.c_int(0), C.c_int(0)
        for cfg in range(ncfg):
            lib.y5_conv_cfg_info(cfg, C.byref(bm), C.byref(bn), C.byref(kb))
            if (bn.value >= 2 * Npad and bn.value > 32) or 57 <= cfg <= 60:
                continue
            d = mk(cfg)
            if lib.y5_conv2d_time(C.byref(d), *ptrs, 5, st, C.byref(ms)) == 0 and ms.value < best_ms:
                best, best_ms = cfg, ms.value
        d = mk(best)

        def launch():
            _lib.check(lib.y5_conv2d_fwd(C.byref(d), *ptrs, st), lib)

        def timed(prep):
            ts = []
            for _ in range(a.reps):
                prep()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            return ts[len(ts) // 2]

        evict = lambda: big_b.copy_(big_a)  # noqa: E731
        sink = []

        def in_warm():
            evict()
            sink.append(x.float().abs().amax())

        def both_warm():
            evict()
            sink.append(x.float().abs().amax())
            sink.append(wp.float().abs().amax())

        launch()
        row = {"layer": name, "cfg": best, "warm_us": round(timed(lambda: launch()), 1), "cold_us": round(timed(evict), 1),
               "input_warm_us": round(timed(in_warm), 1), "both_warm_us": round(timed(both_warm), 1), "filter_mbytes": round(wp.numel() * 2 / 1e6, 2),
               "input_mbytes": round(x.numel() * 2 / 1e6, 1)}
        row["filter_prefetch_gain_us"] = round(row["input_warm_us"] - row["both_warm_us"], 1)
        print(json.dumps(row), flush=True)
        out.append(row)
        sink.clear()
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
