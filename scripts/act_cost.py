"""Kernel-experiment helper: what does the SiLU epilogue cost?  The P4 pointwise layers of yolov5s (bs 64, 40 x 40, 256 -> 256 and 512 -> 256: K = 256 / 512) and
two 3x3 layers on the implicit-GEMM ids the plan uses, timed with act = 1 and act = 0 (y5_conv2d_time, 20 launches each)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yolov5_amd import _lib  # noqa: E402
from yolov5_amd.packing import pack_conv_weight  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
B = 64
for name, H, C1, C2, k, s, cfgs in (("6.cv3 1x1 256->256 @40", 40, 256, 256, 1, 1, (43, 39, 37, 94)), ("13.cv1+cv2 1x1 512->256 @40", 40, 512, 256, 1, 1, (43, 39)),
                                   ("4.cv3 1x1 128->128 @80", 80, 128, 128, 1, 1, (84, 43)), ("5.Conv 3x3s2 128->256 @80", 80, 128, 256, 3, 2, (39, 43)),
                                   ("8.cv3 1x1 512->512 @20", 20, 512, 512, 1, 1, (39,))):
    p = k // 2
    OH = (H + 2 * p - k) // s + 1
    x = torch.randn((B, H, H, C1), device=dev, dtype=torch.float16)
    w = torch.randn((C2, C1, k, k), device=dev) * 0.05
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, torch.zeros(C2, device=dev), torch.float16)
    y = torch.zeros((B, OH, OH, C2), device=dev, dtype=torch.float16)
    flop = 2.0 * B * OH * OH * C2 * C1 * k * k
    out = []
    for cfg in cfgs:
        t = {}
        for act in (1, 0, 1, 0):
            d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=H, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=act, Kpad=Kpad,
                              Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=0)
            ms = C.c_float(0)
            rc = lib.y5_conv2d_time(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), None, C.c_void_p(y.data_ptr()), None, 20, st,
                                    C.byref(ms))
            if rc == 0:
                t[act] = min(t.get(act, 1e9), ms.value)
        if t:
            out.append(f"cfg {cfg}: SiLU {t[1] * 1e3:.1f} us ({flop / t[1] / 1e9:.0f} TF), no activation {t[0] * 1e3:.1f} us ({flop / t[0] / 1e9:.0f} TF)")
    print(f"{name:32s} " + " | ".join(out))
