"""CPU: y5_conv2d_wgrad (yolov5_amd/csrc/wgrad.hip) on the HIP emulator vs torch autograd's conv weight gradient."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib
from yolov5_amd.packing import round_up


def run_wgrad(lib, x_nchw, dz_nchw, k, s, p, splits, ldx_extra=8, ldz_extra=8, aligned_fn=aligned, ptr_fn=ptr, stream=None, cfg=-1):
    B, C1, H, W = x_nchw.shape
    C2 = dz_nchw.shape[1]
    kh, kw = k; sh, sw = s; ph, pw = p
    OH, OW = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    assert dz_nchw.shape[2:] == (OH, OW)
    ldx, ldz = C1 + ldx_extra, C2 + ldz_extra
    x = aligned_fn((B, H, W, ldx), np.float16, 5.0); x[..., :C1] = x_nchw.permute(0, 2, 3, 1).numpy()
    dz = aligned_fn((B, OH, OW, ldz), np.float16, 5.0); dz[..., :C2] = dz_nchw.permute(0, 2, 3, 1).numpy()
    K = kh * kw * C1
    Kpad, Npad = round_up(K, 64), round_up(C2, 32)
    dw = aligned_fn((Npad, Kpad), np.float32, 0.0)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=ldx, OH=OH, OW=OW, C2=C2, ldy=ldz, KH=kh, KW=kw, SH=sh, SW=sw,
                      PH=ph, PW=pw, act=0, Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=splits)
    rc = lib.y5_conv2d_wgrad(C.byref(d), ptr_fn(x), ptr_fn(dz), ldz, ptr_fn(dw), stream)
    assert rc == 0, lib.y5_last_error()
    return dw, K, Kpad


CASES = [
    # B, H, W, C1, C2, k, s, p, splits
    (2, 6, 7, 32, 32, (1, 1), (1, 1), (0, 0), 0),
    (2, 9, 8, 32, 40, (3, 3), (1, 1), (1, 1), 3),     # C1=32: a 64-wide k tile spans two taps; C2 tail
    (1, 12, 12, 64, 64, (3, 3), (2, 2), (1, 1), 2),
    (2, 8, 8, 16, 72, (3, 3), (1, 1), (1, 1), 0),     # K = 144 -> 3 k tiles with a tail, two n tiles
    (1, 16, 8, 8, 32, (6, 3), (2, 1), (2, 1), 0),     # the stem's paired-pixel view (NHWC4 x 2 = 8 channels)
    (3, 5, 5, 128, 64, (1, 1), (1, 1), (0, 0), 4),
    (2, 6, 6, 32, 136, (3, 3), (1, 1), (1, 1), 2),    # 128 x 128 block tile (2 x 2 accumulators per wave), n tail
    (2, 7, 5, 64, 128, (1, 1), (1, 1), (0, 0), 0),    # 128 x 64 block tile
    (2, 6, 6, 128, 136, (1, 1), (1, 1), (0, 0), 3),   # pointwise (linear staging) on the 128 x 128 tile, n tail, 3 pixel splits
]


@pytest.mark.parametrize("case", CASES)
def test_emu_wgrad_matches_torch(case):
    B, H, W, C1, C2, k, s, p, splits = case
    lib = emu()
    x = torch.from_numpy(detgen.uniform((B, C1, H, W), -1, 1, name="wx")).half()
    OH, OW = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="wdz")).half()
    dw, K, Kpad = run_wgrad(lib, x, dz, k, s, p, splits)
    w = torch.zeros((C2, C1, k[0], k[1]), requires_grad=True)
    F.conv2d(x.float(), w, None, s, p).backward(dz.float())
    ref = w.grad.permute(0, 2, 3, 1).reshape(C2, K).numpy()   # k = (kh, kw, c)
    np.testing.assert_allclose(dw[:C2, :K], ref, rtol=2e-3, atol=2e-3)
    assert np.all(dw[C2:] == 0) and np.all(dw[:, K:] == 0)


# the patch-staged 3x3 family (csrc/wgrad3.h; cfg 3 = forced, 1 = the general gather kernel on the same geometry)
CASES3 = [
    # B, H, W, C1, C2, s, splits
    (1, 5, 40, 32, 32, 1, 0),      # NT 1 CT 1; OW = 40: three segments per row, the last one half empty
    (2, 7, 20, 32, 64, 1, 3),      # NT 2; OW = 20: 16 + 4
    (1, 9, 41, 32, 64, 2, 2),      # stride 2, odd sizes: OW = 21, right / bottom border taps
    (2, 6, 18, 64, 64, 1, 0),      # NT 2 CT 2
    (1, 8, 34, 64, 128, 2, 0),     # NT 4 CT 2 stride 2 (the 33-pixel patch rows)
    (1, 6, 16, 128, 160, 1, 2),    # two c tiles x two n tiles (n tail), OW = 16 exactly
    (2, 4, 6, 24, 40, 1, 0),       # C1 = 24: channel tail inside the single c tile; OW < 16
    (1, 3, 70, 40, 32, 2, 5),      # CT 2 with a c tail (40 of 64), OW = 35, more splits than needed
]


@pytest.mark.parametrize("case", CASES3)
@pytest.mark.parametrize("cfg", [3, 1, 321, -1, 111, 112])
def test_emu_wgrad_k3_families_match_torch(case, cfg):
    B, H, W, C1, C2, s, splits = case
    lib = emu()
    x = torch.from_numpy(detgen.uniform((B, C1, H, W), -1, 1, name="w3x")).half()
    OH, OW = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="w3dz")).half()
    dw, K, Kpad = run_wgrad(lib, x, dz, (3, 3), (s, s), (1, 1), splits, cfg=cfg)
    w = torch.zeros((C2, C1, 3, 3), requires_grad=True)
    F.conv2d(x.float(), w, None, s, 1).backward(dz.float())
    ref = w.grad.permute(0, 2, 3, 1).reshape(C2, K).numpy()
    np.testing.assert_allclose(dw[:C2, :K], ref, rtol=2e-3, atol=2e-3)
    assert np.all(dw[C2:] == 0) and np.all(dw[:, K:] == 0)


@pytest.mark.parametrize("B,H,W,C2,splits,cfg", [(1, 24, 40, 32, 0, 6), (2, 10, 17, 32, 3, 6), (1, 12, 36, 24, 2, -1), (2, 8, 16, 40, 0, 6), (1, 24, 40, 32, 0, 1)])
def test_emu_wgrad_stem_kernel_matches_torch(B, H, W, C2, splits, cfg):
    """cfg 6 (csrc/wgrad3.h y5_conv_wgrad_stem_kernel): 0.Conv's weight gradient on the paired-pixel view -- k(6,3) s(2,1) p(2,1), 8 channels, pixels
    contiguous (ldx = 8) -- against torch's conv2d weight gradient of the same geometry; -1 = the library's automatic choice (the stem kernel), 1 = the
    general gather kernel on the same buffers."""
    lib = emu()
    x = torch.from_numpy(detgen.uniform((B, 8, H, W), -1, 1, name="stx")).half()
    OH, OW = (H + 4 - 6) // 2 + 1, W
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="stdz")).half()
    dw, K, Kpad = run_wgrad(lib, x, dz, (6, 3), (2, 1), (2, 1), splits, ldx_extra=0, cfg=cfg)
    w = torch.zeros((C2, 8, 6, 3), requires_grad=True)
    F.conv2d(x.float(), w, None, (2, 1), (2, 1)).backward(dz.float())
    ref = w.grad.permute(0, 2, 3, 1).reshape(C2, K).numpy()
    np.testing.assert_allclose(dw[:C2, :K], ref, rtol=2e-3, atol=2e-3)
    assert np.all(dw[C2:] == 0) and np.all(dw[:, K:] == 0)


def test_emu_wgrad_cfg3_refuses_other_geometries():
    lib = emu()
    x = torch.zeros((1, 32, 6, 6)).half()
    dz = torch.zeros((1, 32, 6, 6)).half()
    with pytest.raises(AssertionError):
        run_wgrad(lib, x, dz, (1, 1), (1, 1), (0, 0), 0, cfg=3)


@pytest.mark.parametrize("case,cfg", [(CASES[1], -1), (CASES[5], -1), (CASES[8], -1), (CASES[2], 3), (CASES[6], 3), (CASES[6], 341)])
def test_emu_wgrad_deterministic_form_equals_atomic_form(case, cfg):
    """y5_conv2d_wgrad_det: the pixel-range splits park their partial tiles in a workspace and a second launch adds them in
    split order.  Same sums as the atomic form up to fp32 association; on top of a non-zero dW (+=); too small a workspace is refused."""
    B, H, W, C1, C2, k, s, p, splits = case
    lib = emu()
    x = torch.from_numpy(detgen.uniform((B, C1, H, W), -1, 1, name="wx")).half()
    OH, OW = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
    dz = torch.from_numpy(detgen.uniform((B, C2, OH, OW), -1, 1, name="wdz")).half()
    ref, K, Kpad = run_wgrad(lib, x, dz, k, s, p, splits, cfg=cfg)
    Npad = round_up(C2, 32)
    ldx, ldz = C1 + 8, C2 + 8
    xa = aligned((B, H, W, ldx), np.float16, 5.0); xa[..., :C1] = x.permute(0, 2, 3, 1).numpy()
    za = aligned((B, OH, OW, ldz), np.float16, 5.0); za[..., :C2] = dz.permute(0, 2, 3, 1).numpy()
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=ldx, OH=OH, OW=OW, C2=C2, ldy=ldz, KH=k[0], KW=k[1], SH=s[0], SW=s[1],
                      PH=p[0], PW=p[1], act=0, Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=splits)
    need = lib.y5_conv2d_wgrad_ws_bytes(C.byref(d), ldz)
    assert need > 0 and need % (Npad * Kpad * 4) == 0
    ws = aligned((need // 4,), np.float32, 123.0)                      # garbage: every slab element that is read must have been written
    outs = []
    for _ in range(2):
        dw = aligned((Npad, Kpad), np.float32, 0.0)
        dw[:C2, :K] = 1.5                                              # accumulates on top of what is there
        rc = lib.y5_conv2d_wgrad_det(C.byref(d), ptr(xa), ptr(za), ldz, ptr(dw), ptr(ws), need, None)
        assert rc == 0, lib.y5_last_error()
        outs.append(dw.copy())
    assert np.array_equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0][:C2, :K] - 1.5, ref[:C2, :K], rtol=1e-5, atol=1e-4)
    assert np.all(outs[0][C2:] == 0) and np.all(outs[0][:, K:] == 0)
    assert lib.y5_conv2d_wgrad_det(C.byref(d), ptr(xa), ptr(za), ldz, ptr(outs[0]), ptr(ws), need - 16, None) != 0
