"""Diagnostic (GPU): every conv launch the training engine autotunes is re-run with every applicable tile configuration on
the engine's own buffers and compared with the heuristic configuration's output.  Prints the configurations that differ."""
import argparse
import copy
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import detgen, yolo_oracle as yo  # noqa: E402  (synthetic weights/targets only)
from yolov5_amd import _lib, engine, train_engine  # noqa: E402


class _Ptr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f2", "data": (ptr, False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov5n")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from yolov5_amd.yolo import DetectionModel
    from yolov5_amd.loss import ComputeLoss

    captured = []
    orig = engine.autotune_conv

    def spy(lib, d, ptrs, st):
        dd = copy.copy(d)
        captured.append((dd, tuple(None if (p is None or p.value is None) else p.value for p in ptrs)))
        return orig(lib, d, ptrs, st)

    train_engine.autotune_conv = spy
    cfg = yo.model_cfg(a.model)
    m = DetectionModel(a.model + ".yaml")
    m.load_state_dict(yo.det_state_dict(cfg, 0, fused=False))
    m.hyp = dict(yo.HYP_SCRATCH_LOW)
    m = m.to(dev).train()
    x = torch.rand((a.batch, 3, a.size, a.size), device=dev).half()
    t = torch.from_numpy(detgen.synth_targets(a.batch, 6, seed=7)).to(dev)
    loss, _ = ComputeLoss(m)(m(x), t)
    loss.backward()
    torch.cuda.synchronize()
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nbad = 0
    for d, ptrs in captured:
        placed = d.out_mul_h != 0
        npix = d.B * (d.out_H * d.out_W if placed else d.OH * d.OW)
        n = (npix - 1) * d.ldy + d.C2
        out = torch.as_tensor(_Ptr(ptrs[4], n), device=dev)
        vp = tuple(C.c_void_p(p) if p else None for p in ptrs)
        d.cfg = -1
        sl = torch.as_strided(out, (npix, d.C2), (d.ldy, 1))
        if ptrs[3] is None:
            sl.fill_(777.0)
        _lib.check(lib.y5_conv2d_fwd(C.byref(d), *vp, st), lib)
        ref = out.clone()
        desc = f"B{d.B} {d.H}x{d.W} C1={d.C1} C2={d.C2} k{d.KH}x{d.KW} s{d.SH} p{d.PH} act{d.act} res={ptrs[3] is not None} Npad={d.Npad} Kpad={d.Kpad} ldx={d.ldx} ldy={d.ldy} placed={placed}"
        for c in range(lib.y5_conv_num_cfgs()):
            d.cfg = c
            if ptrs[3] is None:
                sl.fill_(777.0)
            rc = lib.y5_conv2d_fwd(C.byref(d), *vp, st)
            if rc:
                continue
            torch.cuda.synchronize()
            diff = (out.float() - ref.float()).abs()
            mx = float(diff.max())
            scale = float(ref.float().abs().max()) + 1e-6
            rel = float(diff.double().norm() / (ref.double().norm() + 1e-30))
            if not (mx <= 2e-2 * scale and rel < 1.5e-3):
                nbad += 1
                print(f"MISMATCH cfg {c}: max|d|={mx:.4g} relL2={rel:.3g} (scale {scale:.3g}, {int((diff > 2e-2 * scale).sum())} elems)  {desc}")
    print(f"checked {len(captured)} conv launches, {nbad} mismatching (cfg, layer) pairs")


if __name__ == "__main__":
    main()
