"""Kernel-experiment helper (not a test): y5_bottleneck_cv3_fwd at C = 128 over a grid of shape variations, error location map for the failing ones."""
import ctypes as C
import itertools
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from yolov5_amd import _lib  # noqa: E402
from yolov5_amd.packing import pack_conv_weight  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
Cc = 128


def run(B, H, W, add, c3, ldx, ld2, ldo, act3, mb, seed=0, off2=None, offx=None, plain=False):
    off2 = ld2 - Cc if off2 is None else off2
    offx = ldx - Cc if offx is None else offx
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn((Cc, Cc, 1, 1), generator=g) * (2.0 / Cc) ** 0.5
    w2 = torch.randn((Cc, Cc, 3, 3), generator=g) * (2.0 / (9 * Cc)) ** 0.5
    w3 = torch.randn((c3, 2 * Cc, 1, 1), generator=g) * (2.0 / (2 * Cc)) ** 0.5
    b1, b2, b3 = torch.randn(Cc, generator=g) * 0.3, torch.randn(Cc, generator=g) * 0.3, torch.randn(c3, generator=g) * 0.3
    w1p, b1p, _, K1, _ = pack_conv_weight(w1, b1, torch.float16)
    w2p, b2p, _, K2, _ = pack_conv_weight(w2, b2, torch.float16)
    w3p, b3p, _, K3, N3 = pack_conv_weight(w3, b3, torch.float16)
    w1p, b1p, w2p, b2p, w3p, b3p = (t.to(dev) for t in (w1p, b1p, w2p, b2p, w3p, b3p))
    xbuf = torch.randn((B, H, W, ldx), generator=g).half().to(dev)
    y2buf = torch.randn((B, H, W, ld2), generator=g).half().to(dev)
    obuf = torch.full((B, H, W, ldo), 7.0, dtype=torch.float16, device=dev)
    st = _lib.stream(dev)
    vp = lambda t, off=0: C.c_void_p(t.data_ptr() + off)  # noqa: E731
    rc = lib.y5_bottleneck_cv3_fwd(vp(xbuf, offx * 2), ldx, vp(w1p), vp(b1p), K1, vp(w2p), vp(b2p), K2, vp(y2buf, off2 * 2), ld2, vp(w3p), vp(b3p), K3,
                                   c3, act3, vp(obuf), ldo, B, H, W, Cc, int(add), mb, st)
    _lib.check(rc, lib)
    torch.cuda.synchronize()
    xf = xbuf[..., offx:offx + Cc].float().permute(0, 3, 1, 2)
    y2f = y2buf[..., off2:off2 + Cc].float().permute(0, 3, 1, 2)
    t = F.silu(F.conv2d(xf, w1.half().float().to(dev), b1.to(dev))).half().float()
    m = F.silu(F.conv2d(t, w2.half().float().to(dev), b2.to(dev), padding=1)).half().float()
    if add:
        m = (m + xf).half().float()
    o = F.conv2d(torch.cat((m, y2f), 1), w3.half().float().to(dev), b3.to(dev))
    ref = (F.silu(o) if act3 & 1 else o).half().float().permute(0, 2, 3, 1)
    got = obuf[..., :c3].float()
    err = (got - ref).abs()
    bad = err > 2e-2
    tag = f"off2={off2} offx={offx} B{B} {H}x{W} add{int(add)} c3={c3} ldx{ldx} ld2={ld2} ldo{ldo} act{act3} mb{mb} N3={N3} K3={K3}"
    if not bool(bad.any()):
        print("ok  ", tag, f"max {float(err.max()):.4f}")
        return
    print("FAIL", tag, f"max {float(err.max()):.3f} bad {int(bad.sum())} of {bad.numel()}")
    px = bad.any(-1)
    for b in range(B):
        if bool(px[b].any()):
            rows = ["".join("#" if bool(px[b, h, w]) else "." for w in range(W)) for h in range(H)]
            print(f" image {b}:\n   " + "\n   ".join(rows))
    ch = bad.any(0).any(0).any(0)
    print(" bad channels:", "".join("#" if bool(c) else "." for c in ch))


base = dict(B=3, H=23, W=37, add=False, c3=248, ldx=128, ld2=136, ldo=264, act3=0, mb=8)
for rep in range(3):
    run(**base, seed=rep)
    run(**dict(base, ld2=160), seed=rep)
    run(**dict(base, ld2=144, add=True, act3=1, mb=0), seed=rep)
    run(**dict(base, ld2=136, ldx=136, B=5, H=41, W=39, mb=0), seed=rep)
