"""Stream-K configurations (57..60) vs the best plain configuration on the deep-layer shapes of yolov5s bs=64: correctness (vs the plain
result) and time per launch."""
import ctypes as C, os, sys
os.environ["Y5_EXPERIMENTAL"] = "streamk"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5_amd import _lib
from yolov5_amd.engine import SK_CFGS, autotune_conv, _ensure_sk_workspace, _HipBackend
from yolov5_amd.packing import pack_conv_weight

dev = torch.device("cuda:0")
lib = _lib.lib()
st = _lib.stream(dev)
be = _HipBackend(dev)
_ensure_sk_workspace(be, lib, st)
vp = lambda t: C.c_void_p(t.data_ptr())
B = 64
shapes = [("7.Conv 3x3s2 256->512 @40", 40, 256, 512, 3, 2), ("8.m.cv2 3x3 256->256 @20", 20, 256, 256, 3, 1), ("9.cv2 1x1 1024->512 @20", 20, 1024, 512, 1, 1),
          ("6.m.cv2 3x3 128->128 @40", 40, 128, 128, 3, 1), ("5.Conv 3x3s2 128->256 @80", 80, 128, 256, 3, 2), ("21.Conv 3x3s2 256->256 @40", 40, 256, 256, 3, 2),
          ("8.cv3 1x1 512->512 @20", 20, 512, 512, 1, 1), ("13.cv1+cv2 1x1 512->256 @40", 40, 512, 256, 1, 1)]
for name, HW, C1, C2, k, s in shapes:
    torch.manual_seed(0)
    w = torch.randn(C2, C1, k, k) * (2.0 / (C1 * k * k)) ** 0.5
    b = torch.randn(C2) * 0.1
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    wp, bp = wp.to(dev), bp.to(dev)
    x = torch.randn(B, HW, HW, C1, device=dev).half()
    p = k // 2
    OH = (HW + 2 * p - k) // s + 1
    y = torch.empty(B, OH, OH, C2, device=dev, dtype=torch.float16)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=HW, W=HW, C1=C1, ldx=C1, OH=OH, OW=OH, C2=C2, ldy=C2, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=1,
                      Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=-1, max_blocks=0)
    ptrs = (vp(x), vp(wp), vp(bp), None, vp(y), None)
    best = autotune_conv(lib, d, ptrs, st, exclude=SK_CFGS)
    ms = C.c_float(0)
    d.cfg = best
    _lib.check(lib.y5_conv2d_time(C.byref(d), *ptrs, 20, st, C.byref(ms)), lib)
    ref = y.clone()
    line = f"{name}: plain cfg {best} {ms.value*1e3:.1f} us |"
    for cfg in sorted(SK_CFGS):
        d.cfg = cfg
        y.zero_()
        rc = lib.y5_conv2d_time(C.byref(d), *ptrs, 20, st, C.byref(ms))
        if rc != 0:
            line += f" sk{cfg} n/a |"
            continue
        torch.cuda.synchronize()
        err = (y.float() - ref.float()).abs().max().item()
        line += f" sk{cfg} {ms.value*1e3:.1f} us (err {err:.3f}) |"
    print(line, flush=True)
