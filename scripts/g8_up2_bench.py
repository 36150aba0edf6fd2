#!/usr/bin/env python
"""Round 6: the virtual Upsample + Concat layers (13.C3.cv1+cv2, 17.C3.cv1+cv2 of yolov5s at bs 64) on ids 88 / 89 (conv_igemm.h UP2) and 95 / 96 (conv_g8.h UP2):
interleaved rounds in one process, each arm checked against torch on the materialised concat first."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

lib = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream(dev)
vp = lambda t: C.c_void_p(t.data_ptr())
for name, B, H, c_up, c_hi, C2, split in [("13.C3.cv1+cv2 @40", 64, 40, 256, 256, 256, 128), ("17.C3.cv1+cv2 @80", 64, 80, 128, 128, 128, 64)]:
    torch.manual_seed(0)
    lo = torch.randn((B, H // 2, H // 2, c_up), device=dev).half()
    x = torch.randn((B, H, H, c_up + c_hi), device=dev).half()
    x[..., :c_up] = float("nan")
    w = (torch.randn((C2, c_up + c_hi, 1, 1), device=dev) * (2.0 / (c_up + c_hi)) ** 0.5).half().float()
    b = torch.randn(C2, device=dev) * 0.3
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    cat = torch.cat((F.interpolate(lo.permute(0, 3, 1, 2).float(), scale_factor=2, mode="nearest"), x[..., c_up:].permute(0, 3, 1, 2).float()), 1)
    ref = F.silu(F.conv2d(cat, w, b)).permute(0, 2, 3, 1)
    y = torch.empty((B, H, H, split), device=dev, dtype=torch.float16)
    y2 = torch.empty((B, H, H, C2 - split), device=dev, dtype=torch.float16)
    flop = 2.0 * B * H * H * C2 * (c_up + c_hi)

    def desc(cfg):
        return _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=c_up + c_hi, ldx=c_up + c_hi, OH=H, OW=H, C2=C2, ldy=split, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0, act=1,
                             Kpad=Kpad, Npad=Npad, ldr=0, ld2=C2 - split, cfg=cfg, max_blocks=0, split_n=split, up_c=c_up, ld_up=c_up)

    arms = []
    for cfg in (88, 89, 95, 96):
        d = desc(cfg)
        y.fill_(0); y2.fill_(0)
        rc = lib.y5_conv2d_fwd(C.byref(d), vp(x), vp(wp), vp(bp), vp(lo), vp(y), vp(y2), st)
        torch.cuda.synchronize()
        if rc != 0:
            print(cfg, "rejected:", lib.y5_last_error()); continue
        err = (torch.cat((y, y2), -1).float() - ref).abs().max().item()
        assert err < 3e-2, (cfg, err)
        arms.append(cfg)
    times = {c: [] for c in arms}
    ms = C.c_float(0)
    for _ in range(5):
        for c in arms:
            d = desc(c)
            _lib.check(lib.y5_conv2d_time(C.byref(d), vp(x), vp(wp), vp(bp), vp(lo), vp(y), vp(y2), 20, st, C.byref(ms)), lib)
            times[c].append(ms.value * 1e3)
    print(f"{name:22s} " + "  ".join(f"[{c}] {sorted(times[c])[2]:6.1f} us {flop / sorted(times[c])[2] / 1e6:5.0f} TF" for c in arms), flush=True)
