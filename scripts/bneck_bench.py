"""Fused Bottleneck (y5_bottleneck_fwd) vs the two launches it replaces (1x1 pointwise + 3x3 with residual epilogue), yolov5s bs=64 shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5_amd import _lib
from yolov5_amd.engine import autotune_conv
from yolov5_amd.packing import pack_conv_weight

dev = torch.device("cuda:0")
lib = _lib.lib()
st = _lib.stream(dev)
vp = lambda t: C.c_void_p(t.data_ptr())

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

import os as _os
SHAPES = ((128, 40, 64),) if _os.environ.get("BNECK_ONLY128") else ((32, 160, 64), (64, 80, 64), (128, 40, 64))
for (Cc, HW, B) in SHAPES:
    torch.manual_seed(0)
    w1 = torch.randn(Cc, Cc, 1, 1) * (2.0 / Cc) ** 0.5; b1 = torch.randn(Cc) * 0.1
    w2 = torch.randn(Cc, Cc, 3, 3) * (2.0 / (9 * Cc)) ** 0.5; b2 = torch.randn(Cc) * 0.1
    w1p, b1p, _, K1, N1 = pack_conv_weight(w1, b1, torch.float16)
    w2p, b2p, _, K2, N2 = pack_conv_weight(w2, b2, torch.float16)
    w1p, b1p, w2p, b2p = (t.to(dev) for t in (w1p, b1p, w2p, b2p))
    cat = torch.randn(B, HW, HW, 2 * Cc, device=dev).half()      # x = cat[..., :C] (a slice, like the C3 concat buffer)
    tmp = torch.empty(B, HW, HW, Cc, device=dev, dtype=torch.float16)
    out = torch.empty(B, HW, HW, Cc, device=dev, dtype=torch.float16)
    fused = lambda: _lib.check(lib.y5_bottleneck_fwd(vp(cat), 2 * Cc, vp(w1p), vp(b1p), K1, vp(w2p), vp(b2p), K2, vp(out), Cc, B, HW, HW, Cc, 1, 0, st), lib)
    d1 = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=HW, W=HW, C1=Cc, ldx=2 * Cc, OH=HW, OW=HW, C2=Cc, ldy=Cc, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0, act=1,
                       Kpad=K1, Npad=N1, ldr=0, ld2=0, cfg=-1, max_blocks=0)
    p1 = (vp(cat), vp(w1p), vp(b1p), None, vp(tmp), None)
    d1.cfg = autotune_conv(lib, d1, p1, st)
    d2 = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=HW, W=HW, C1=Cc, ldx=Cc, OH=HW, OW=HW, C2=Cc, ldy=2 * Cc, KH=3, KW=3, SH=1, SW=1, PH=1, PW=1, act=1,
                       Kpad=K2, Npad=N2, ldr=2 * Cc, ld2=0, cfg=-1, max_blocks=0)
    cat2 = cat.clone()
    p2 = (vp(tmp), vp(w2p), vp(b2p), vp(cat2), vp(cat2), None)
    scratch = torch.empty_like(cat)
    d2.cfg = autotune_conv(lib, d2, (vp(tmp), vp(w2p), vp(b2p), vp(cat), vp(scratch), None), st)
    two = lambda: (_lib.check(lib.y5_conv2d_fwd(C.byref(d1), *p1, st), lib), _lib.check(lib.y5_conv2d_fwd(C.byref(d2), *p2, st), lib))
    # numerics: fused vs two-op on the same input
    fused(); 
    _lib.check(lib.y5_conv2d_fwd(C.byref(d1), *p1, st), lib)
    ref = torch.empty_like(out)
    d2o = _lib.ConvDesc.from_buffer_copy(d2); d2o.ldy = Cc
    _lib.check(lib.y5_conv2d_fwd(C.byref(d2o), vp(tmp), vp(w2p), vp(b2p), vp(cat), vp(ref), None, st), lib)
    torch.cuda.synchronize()
    err = (out.float() - ref.float()).abs().max().item()
    mbs = (0, 18 << 16, 4 << 16, 5 << 16) if Cc == 128 else (0,) + tuple(g | (S << 16) for S in ((1, 2, 3) if Cc == 32 else (1,)) for g in (256, 512, 768))
    for mb in mbs:
        f2 = lambda: _lib.check(lib.y5_bottleneck_fwd(vp(cat), 2 * Cc, vp(w1p), vp(b1p), K1, vp(w2p), vp(b2p), K2, vp(out), Cc, B, HW, HW, Cc, 1, mb, st), lib)
        print(f"C={Cc} {HW}^2 bs={B}: fused (stages {mb >> 16}, grid cap {mb & 0xffff}) {timeit(f2):.1f} us")
    print(f"C={Cc} {HW}^2 bs={B}: two launches (cfg {d1.cfg} + {d2.cfg}) {timeit(two):.1f} us; max|fused - two-op| = {err:.4f}")
