#!/usr/bin/env python
"""The fused backbone front (y5_conv_front_fwd: 0.Conv + 1.Conv + 2.C3.cv1+cv2 in one launch, csrc/conv_front.h) against the two launches it
replaces (y5_conv_stem_fwd + y5_conv_k3pw_fwd) on one MI355X: same outputs (fp16 tolerance: the bias enters the accumulation chain first instead
of last) and HIP-event timing of both forms at yolov5s bs=64 640^2."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight, pack_stem_weight


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--blocks", default="0")
    a = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator().manual_seed(0)
    B, H = a.batch, a.imgsz
    x = torch.rand((B, 3, H, H), generator=g).half().to(dev)
    w0, b0 = torch.randn((32, 3, 6, 6), generator=g) * (2 / 108) ** 0.5, torch.randn(32, generator=g) * 0.3
    w1, b1 = torch.randn((64, 32, 3, 3), generator=g) * (2 / 288) ** 0.5, torch.randn(64, generator=g) * 0.3
    w2, b2 = torch.randn((64, 64, 1, 1), generator=g) * (2 / 64) ** 0.5, torch.randn(64, generator=g) * 0.3
    w0p, b0p, _ = pack_stem_weight(w0.to(dev), b0.to(dev))
    w1p, b1p, _, K1, N1 = pack_conv_weight(w1.to(dev), b1.to(dev), torch.float16)
    w2p, b2p, _, K2, N2 = pack_conv_weight(w2.to(dev), b2.to(dev), torch.float16)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    OH0, OH1 = H // 2, H // 4
    s0 = torch.empty((B, OH0, OH0, 32), dtype=torch.float16, device=dev)
    ya, yb = (torch.zeros((B, OH1, OH1, 32), dtype=torch.float16, device=dev) for _ in range(2))
    fa, fb = (torch.zeros((B, OH1, OH1, 32), dtype=torch.float16, device=dev) for _ in range(2))
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=OH0, W=OH0, C1=32, ldx=32, OH=OH1, OW=OH1, C2=64, ldy=64, KH=3, KW=3, SH=2, SW=2, PH=1, PW=1, act=1,
                      Kpad=K1, Npad=N1, ldr=0, ld2=0, cfg=81, max_blocks=0)

    def two():
        _lib.check(lib.y5_conv_stem_fwd(p(x), B, H, H, p(w0p), p(b0p), 32, 32, p(s0), 32, 0, st), lib)
        _lib.check(lib.y5_conv_k3pw_fwd(C.byref(d), p(s0), p(w1p), p(b1p), p(w2p), p(b2p), 64, N2, K2, 1, p(ya), 32, p(yb), 32, 32, st), lib)

    def front(mb=0):
        _lib.check(lib.y5_conv_front_fwd(p(x), B, H, H, p(w0p), p(b0p), 32, p(w1p), p(b1p), 64, N1, K1, 1, p(w2p), p(b2p), 64, N2, K2, 1, p(fa), 32, p(fb), 32,
                                         32, mb, st), lib)

    two()
    front()
    torch.cuda.synchronize()
    da, db = (fa.float() - ya.float()).abs(), (fb.float() - yb.float()).abs()
    res = {"max_abs_diff": [float(da.max()), float(db.max())], "mean_abs_diff": [float(da.mean()), float(db.mean())], "ref_abs_mean": float(ya.float().abs().mean()),
           "mismatch_frac_gt_1e-2": float(((da > 1e-2 + 1e-2 * ya.float().abs()).float().mean()))}

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3

    res["two_launch_us"] = round(timed(two), 1)
    for mb in [int(v) for v in a.blocks.split(",")]:
        res[f"front_us_blocks{mb}"] = round(timed(lambda: front(mb)), 1)
    print(json.dumps(res))
    if hasattr(lib, "y5_front_dbg_read") or os.environ.get("Y5_LIB_PATH"):
        try:
            lib.y5_front_dbg_read.restype = C.c_int
            lib.y5_front_dbg_read.argtypes = [C.c_void_p]
            front(0)
            torch.cuda.synchronize()
            dbg = (C.c_ulonglong * 64)()
            assert lib.y5_front_dbg_read(dbg) == 0
            for wv in range(8):
                o = dbg[wv * 8: wv * 8 + 7]
                n = max(o[6], 1)
                print(f"   wave {wv}: tiles {o[6]}  cycles per tile: input wait+barrier {o[0] / n:.0f}  stem {o[1] / n:.0f}  patch barrier {o[2] / n:.0f}  "
                      f"input issue {o[3] / n:.0f}  3x3 {o[4] / n:.0f}  1x1+stores {o[5] / n:.0f}  total {sum(o[:6]) / n:.0f}")
        except AttributeError:
            pass


if __name__ == "__main__":
    main()
