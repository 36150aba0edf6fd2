"""Host-side (one-off, at model load) repacking of convolution filters into the layout the HIP implicit-GEMM
kernel streams: [Npad][Kpad] row-major with k = (kh, kw, c), zero padded (include/yolov5_hip.h: y5_conv2d_fwd)."""
from __future__ import annotations

import torch


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_conv_weight(w: torch.Tensor, bias, dtype: torch.dtype, c1_pad: int | None = None):
    """w (C2, C1, KH, KW), bias (C2,) or None -> (w_packed [Npad, Kpad] `dtype`, bias_f32 [Npad], K, Kpad, Npad).

    `c1_pad`: input channels of the NHWC activation this filter will read (>= C1, zero filter taps for the pad).
    """
    c2, c1, kh, kw = w.shape
    c1p = c1 if c1_pad is None else c1_pad
    bk = 128 // torch.empty((), dtype=dtype).element_size()  # K padded to 128 bytes: valid for every tile config
    wk = torch.zeros((c2, kh, kw, c1p), dtype=torch.float32, device=w.device)
    wk[..., :c1] = w.detach().float().permute(0, 2, 3, 1)
    k = kh * kw * c1p
    kpad, npad = round_up(k, bk), round_up(c2, 32)
    wp = torch.zeros((npad, kpad), dtype=torch.float32, device=w.device)
    wp[:c2, :k] = wk.reshape(c2, k)
    bp = torch.zeros((npad,), dtype=torch.float32, device=w.device)
    if bias is not None:
        bp[:c2] = bias.detach().float()
    return wp.to(dtype).contiguous(), bp.contiguous(), k, kpad, npad


def fuse_conv_bn_weights(w, conv_bias, bn_w, bn_b, bn_mean, bn_var, eps):
    """BN folding for eval: W' = diag(g/sqrt(var+eps)) W, b' = (b_conv - mean) * g/sqrt(var+eps) + beta.

    Same result as the reference's utils/torch_utils.py:224-254 `fuse_conv_and_bn` (fp32)."""
    scale = bn_w.float() / torch.sqrt(bn_var.float() + eps)
    wf = w.float() * scale.view(-1, 1, 1, 1)
    b0 = torch.zeros_like(bn_mean, dtype=torch.float32) if conv_bias is None else conv_bias.float()
    bf = (b0 - bn_mean.float()) * scale + bn_b.float()
    return wf, bf


def pack_stem_weight(w: torch.Tensor, bias):
    """Stem filter (C2, 3, 6, 6) -> [Npad][144] fp16 for y5_conv_stem_fwd: k = (c*6 + kh)*8 + kw, taps kw = 6,7 zero."""
    c2, c1, kh, kw = w.shape
    assert (c1, kh, kw) == (3, 6, 6), w.shape
    npad = round_up(c2, 32)
    wk = torch.zeros((npad, 3, 6, 8), dtype=torch.float32, device=w.device)
    wk[:c2, :, :, :6] = w.detach().float()
    bp = torch.zeros((npad,), dtype=torch.float32, device=w.device)
    if bias is not None:
        bp[:c2] = bias.detach().float()
    return wk.reshape(npad, 144).to(torch.float16).contiguous(), bp.contiguous(), npad
