"""Host-side helpers of the training path: data-gradient of a convolution expressed as (a set of) forward
implicit-GEMM launches on transformed filters.  Replaces autograd's conv input backward under train.py:410.

    dx[b, ih, iw, c1] = sum_{kh,kw,c2} dz[b, oh, ow, c2] * W[c2, c1, kh, kw],   ih = oh*s - p + kh

For stride s the input rows split into s parity classes ih = s*i + r; class r only sees the taps with
(r + p - kh) % s == 0, i.e. a small stride-1 correlation over dz whose output lands on every s-th row (y5_conv_desc
out_mul/out_off placement).  Stride 1 is the single class r = 0 with the flipped filter and padding k-1-p.
"""
from __future__ import annotations

import torch

from . import _lib
from .packing import pack_conv_weight


def _axis_classes(k: int, s: int, p: int, n_in: int):
    """Per parity class r: (r, taps [kernel index per sub-tap a], pad_before, n_out)."""
    out = []
    for r in range(s):
        ds = sorted((r + p - kk) // s for kk in range(k) if (r + p - kk) % s == 0)
        n_out = (n_in - r + s - 1) // s
        if not ds or n_out <= 0:
            out.append((r, [], 0, max(n_out, 0)))
            continue
        assert ds == list(range(ds[0], ds[-1] + 1))
        taps = [r + p - s * d for d in ds]          # kernel index read by sub-tap a = d - ds[0]
        out.append((r, taps, -ds[0], n_out))
    return out


def dgrad_subconvs(w: torch.Tensor, stride, pad, in_hw):
    """w (C2, C1, KH, KW) -> list of dicts describing the forward launches that produce dx (C1 channels) from dz (C2)."""
    c2, c1, kh, kw = w.shape
    (sh, sw), (ph, pw), (H, W) = stride, pad, in_hw
    subs = []
    for rh, th, padh, nh in _axis_classes(kh, sh, ph, H):
        for rw, tw, padw, nw in _axis_classes(kw, sw, pw, W):
            if nh == 0 or nw == 0:
                continue
            if not th or not tw:
                subs.append(dict(rh=rh, rw=rw, nh=nh, nw=nw, empty=True))
                continue
            wsub = w[:, :, th][:, :, :, tw].permute(1, 0, 2, 3).contiguous()  # (C1, C2, len(th), len(tw))
            subs.append(dict(rh=rh, rw=rw, nh=nh, nw=nw, empty=False, w=wsub, k=(len(th), len(tw)), pad=(padh, padw)))
    return subs


class ConvDgrad:
    """dx = dgrad(dz) for one convolution; filters are re-packed from the live weight tensor on every call (they change
    each optimizer step).  dz / dx are NHWC fp16 slices given as (data_ptr, pixel stride)."""

    def __init__(self, lib, B, in_hw, out_hw, c1, c2, k, stride, pad):
        self.lib, self.B, self.in_hw, self.out_hw = lib, B, in_hw, out_hw
        self.c1, self.c2, self.k, self.stride, self.pad = c1, c2, k, stride, pad
        self._keep = []

    def launch(self, w: torch.Tensor, dz_ptr: int, ld_dz: int, dx_ptr: int, ld_dx: int, accumulate: bool, stream, scale_w: float = 1.0):
        import ctypes as C

        self._keep = []
        H, W = self.in_hw
        OH, OW = self.out_hw
        wf = w.detach().float() * scale_w if scale_w != 1.0 else w.detach().float()
        for sub in dgrad_subconvs(wf, self.stride, self.pad, (H, W)):
            if sub["empty"]:
                raise NotImplementedError("dgrad: parity class without taps (k < stride) needs a zero fill")
            wp, bp, K, Kpad, Npad = pack_conv_weight(sub["w"], None, torch.float16)
            self._keep += [wp, bp]
            dense = self.stride == (1, 1)
            d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=self.B, H=OH, W=OW, C1=self.c2, ldx=ld_dz, OH=sub["nh"], OW=sub["nw"], C2=self.c1,
                              ldy=ld_dx, KH=sub["k"][0], KW=sub["k"][1], SH=1, SW=1, PH=sub["pad"][0], PW=sub["pad"][1], act=0,
                              Kpad=Kpad, Npad=Npad, ldr=ld_dx if accumulate else 0, ld2=0, cfg=-1, max_blocks=0,
                              out_mul_h=0 if dense else self.stride[0], out_mul_w=0 if dense else self.stride[1],
                              out_off_h=sub["rh"], out_off_w=sub["rw"], out_H=0 if dense else H, out_W=0 if dense else W)
            rc = self.lib.y5_conv2d_fwd(C.byref(d), C.c_void_p(dz_ptr), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()),
                                        C.c_void_p(dx_ptr) if accumulate else None, C.c_void_p(dx_ptr), None, stream)
            _lib.check(rc, self.lib)
