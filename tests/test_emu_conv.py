"""CPU (no GPU): the real conv kernel source, compiled for the host on the HIP emulator (tests/hipemu), against
torch-CPU conv2d on tiny shapes -- checks tiling, im2col addressing, swizzle, tails, epilogue variants."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen
from tests.hipemu.emu import aligned, emu, ptr
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight

CASES = [
    # B, H, W, C1, C2, k, s, p, act, residual, up2, cfg, max_blocks, dtype
    (1, 6, 7, 32, 32, 1, 1, 0, 1, False, False, -1, 0, "f16"),
    (2, 9, 8, 32, 64, 3, 1, 1, 1, True, False, -1, 0, "f16"),
    (1, 12, 12, 64, 48, 3, 2, 1, 1, False, False, -1, 0, "f16"),   # C2 not multiple of 32 (Npad 64), BK64 heuristic
    (1, 8, 8, 64, 128, 1, 1, 0, 0, False, True, -1, 0, "f16"),     # 2x2 wave tiling + upsampled second store
    (1, 5, 5, 16, 16, 3, 1, 1, 1, True, False, -1, 0, "f16"),      # table mode (C1 % 32 != 0), in-place residual
    (1, 8, 8, 40, 256, 1, 1, 0, 1, False, False, 3, 0, "f16"),     # table mode 1x1, BN=256 tile
    (1, 6, 6, 32, 32, 3, 1, 1, 1, True, False, -1, 0, "f32"),
    (1, 7, 5, 4, 32, 3, 2, 1, 1, False, False, -1, 0, "f32"),      # f32 table mode (C1=4)
    (1, 10, 10, 16, 128, 1, 1, 0, 0, False, False, -1, 0, "f32"),
    # persistent multi-tile walks: 2-3 workgroups own several (M,N) tiles each, next tile prefetched during epilogue
    (3, 13, 11, 32, 96, 3, 1, 1, 1, True, True, 0, 2, "f16"),      # 4 M-tiles x 3 N-tiles of 128x32 on 2 blocks
    (2, 16, 16, 64, 64, 1, 1, 0, 1, False, False, 7, 3, "f16"),    # nk = 1: every iteration is a new tile
    (2, 12, 12, 32, 64, 3, 2, 1, 1, False, False, 1, 1, "f32"),    # single block walks all tiles
    # streaming pointwise kernel (conv_pw.h): ring fill / steady state / drain, partial last workgroup tile, upsampled store
    (7, 8, 16, 32, 32, 1, 1, 0, 1, False, False, 14, 1, "f16"),    # 32->32, 7 tiles per wave on one workgroup, S=4
    (7, 8, 16, 64, 32, 1, 1, 0, 0, False, False, 15, 1, "f16"),
    (7, 8, 16, 64, 64, 1, 1, 0, 1, False, False, 16, 1, "f16"),
    (5, 4, 8, 64, 64, 1, 1, 0, 1, False, True, 17, 2, "f16"),      # 5 wave tiles: partial workgroup tile + up2
    (5, 8, 16, 128, 64, 1, 1, 0, 1, False, False, 18, 1, "f16"),
    (5, 8, 16, 128, 128, 1, 1, 0, 1, False, True, 19, 1, "f16"),
    (3, 8, 16, 128, 120, 1, 1, 0, 1, False, False, 20, 2, "f16"),  # C2 not a multiple of 32 (Npad 128)
    (6, 8, 16, 128, 64, 1, 1, 0, 1, False, False, 21, 1, "f16"),
    (5, 8, 16, 128, 256, 1, 1, 0, 0, False, False, 56, 1, "f16"),  # 128->256 without activation (P3 Detect head), epilogue in two groups
    (3, 8, 16, 128, 248, 1, 1, 0, 1, False, False, 56, 2, "f16"),  # C2 tail inside the second group, partial workgroup tile
    # eight waves per workgroup, one stage per wave
    (5, 8, 16, 128, 128, 1, 1, 0, 1, False, True, 84, 1, "f16"),
    (7, 8, 16, 64, 64, 1, 1, 0, 1, False, False, 85, 1, "f16"),
    (5, 8, 16, 128, 64, 1, 1, 0, 1, False, False, 86, 2, "f16"),
    (3, 8, 16, 128, 248, 1, 1, 0, 0, False, False, 87, 1, "f16"),
    # streaming 3x3 kernel (conv_k3.h): borders on all sides, partial workgroup tile, residual (in place), stride 2
    (2, 8, 16, 32, 32, 3, 1, 1, 1, True, False, 30, 1, "f16"),
    (1, 12, 8, 32, 32, 3, 1, 1, 0, False, False, 33, 2, "f16"),
    (2, 16, 32, 32, 64, 3, 2, 1, 1, False, False, 31, 1, "f16"),
    (1, 24, 16, 32, 64, 3, 2, 1, 1, False, False, 34, 1, "f16"),
    (3, 8, 8, 64, 64, 3, 1, 1, 1, True, False, 32, 1, "f16"),
    (1, 4, 24, 64, 56, 3, 1, 1, 1, False, False, 32, 2, "f16"),
    (3, 8, 8, 64, 64, 3, 1, 1, 1, True, False, 78, 1, "f16"),       # the 64-channel kernel with the filter in registers
    (1, 12, 24, 64, 56, 3, 1, 1, 0, False, False, 79, 2, "f16"),
    (3, 8, 16, 64, 64, 3, 1, 1, 1, True, False, 80, 1, "f16"),      # eight waves per workgroup, one stage each
    (2, 16, 32, 32, 64, 3, 2, 1, 1, False, False, 81, 1, "f16"),
    (3, 8, 16, 32, 32, 3, 1, 1, 1, True, False, 82, 1, "f16"),
    (1, 12, 24, 32, 32, 3, 1, 1, 0, False, False, 83, 2, "f16"),
    # halo-resident 3x3 kernel (conv_h3.h): one / several channel chunks, spatial tiles that do not divide the image, N tiles with a
    # padded tail, residual in place, several tiles per workgroup, 4- and 8-wave layouts
    (2, 9, 8, 32, 64, 3, 1, 1, 1, True, False, 61, 0, "f16"),
    (1, 12, 44, 64, 160, 3, 1, 1, 1, False, False, 61, 2, "f16"),   # 2 row tiles x 2 N tiles (second one 32 real channels) on 2 workgroups
    (2, 7, 50, 64, 64, 3, 1, 1, 0, True, False, 62, 1, "f16"),      # one workgroup walks every tile
    (3, 20, 20, 96, 96, 3, 1, 1, 1, True, False, 63, 2, "f16"),     # whole 20x20 images, 3 chunks
    (1, 13, 40, 64, 128, 3, 1, 1, 1, False, False, 64, 2, "f16"),
    (2, 22, 20, 32, 48, 3, 1, 1, 1, False, False, 65, 0, "f16"),
    (1, 10, 24, 64, 64, 3, 1, 1, 1, True, False, 66, 1, "f16"),
    (1, 12, 44, 64, 160, 3, 1, 1, 1, True, False, 67, 2, "f16"),
    (2, 20, 20, 64, 128, 3, 1, 1, 1, False, False, 68, 0, "f16"),
    (1, 13, 40, 96, 136, 3, 1, 1, 0, False, False, 69, 1, "f16"),
    (2, 11, 23, 64, 96, 3, 1, 1, 1, True, False, 70, 2, "f16"),
    (2, 20, 20, 64, 128, 3, 1, 1, 1, True, False, 71, 1, "f16"),
    (1, 25, 21, 32, 160, 3, 1, 1, 1, False, False, 72, 2, "f16"),
    (2, 13, 40, 96, 128, 3, 1, 1, 1, True, False, 73, 2, "f16"),     # 4-stage ring: 3 chunks, stage index wraps inside and across chunks
    (1, 20, 20, 160, 64, 3, 1, 1, 1, False, False, 74, 1, "f16"),
    (2, 9, 33, 32, 144, 3, 1, 1, 0, True, False, 75, 0, "f16"),
    (2, 13, 40, 96, 128, 3, 1, 1, 1, True, False, 76, 2, "f16"),
    (1, 10, 40, 64, 48, 3, 1, 1, 1, False, False, 77, 1, "f16"),
    # ... at STRIDE 2 (round 5: 3 / 5 / 7 / 18 / 21.Conv): odd and even input sizes, several tiles per image and per workgroup, the small-tile ids 90..92
    (2, 16, 32, 32, 64, 3, 2, 1, 1, False, False, 61, 0, "f16"),
    (1, 23, 41, 64, 128, 3, 2, 1, 1, False, False, 64, 2, "f16"),
    (2, 20, 20, 96, 160, 3, 2, 1, 0, False, False, 73, 1, "f16"),
    (1, 40, 24, 64, 128, 3, 2, 1, 1, False, False, 76, 2, "f16"),
    (2, 32, 64, 64, 128, 3, 2, 1, 1, False, False, 90, 2, "f16"),
    (1, 33, 65, 32, 96, 3, 2, 1, 1, False, False, 91, 1, "f16"),
    (3, 16, 16, 64, 128, 3, 2, 1, 1, False, False, 92, 0, "f16"),
    (2, 13, 40, 96, 128, 3, 1, 1, 1, True, False, 90, 2, "f16"),      # (and the new ids at stride 1)
    (1, 20, 20, 64, 64, 3, 1, 1, 1, False, False, 91, 1, "f16"),
    # K-streamed pointwise kernel (conv_pwk.h, ids 93 / 94): pixel tails, N tails, several N tiles, K from one to many chunks, no activation
    (1, 6, 7, 32, 32, 1, 1, 0, 1, False, False, 93, 0, "f16"),
    (2, 20, 20, 256, 256, 1, 1, 0, 1, False, False, 93, 0, "f16"),
    (3, 9, 11, 96, 248, 1, 1, 0, 0, False, False, 93, 0, "f16"),
    (1, 16, 16, 160, 512, 1, 1, 0, 1, False, False, 93, 0, "f16"),
    (2, 13, 10, 256, 128, 1, 1, 0, 1, False, False, 94, 0, "f16"),
    (1, 8, 8, 64, 320, 1, 1, 0, 1, False, False, 94, 0, "f16"),
    # 256-row / 8-phase implicit GEMM (conv_g8.h, id 95): one K tile, odd / even K-tile counts (the LDS buffer parity flips per output tile), pixel and
    # channel tails, several N tiles, several output tiles per workgroup (the schedule runs across tile boundaries), residual in place, stride 2, no activation
    (1, 6, 7, 64, 32, 1, 1, 0, 1, False, False, 95, 0, "f16"),
    (2, 9, 8, 64, 64, 3, 1, 1, 1, True, False, 95, 0, "f16"),
    (1, 20, 20, 64, 320, 3, 2, 1, 1, False, False, 95, 0, "f16"),
    (3, 12, 12, 128, 264, 1, 1, 0, 0, False, False, 95, 2, "f16"),
    (2, 13, 11, 64, 512, 3, 1, 1, 1, True, False, 95, 1, "f16"),
    (1, 16, 16, 192, 256, 1, 1, 0, 1, False, False, 95, 0, "f16"),
    # ... 256 x 128 tiles (id 96): three resident K tiles, the filter walker one K tile ahead of the activation walker -- K-tile counts 1, 2, 3, 9 (every
    # residue of the buffer rotation across an output-tile boundary), N tiles with a padded tail, several tiles per workgroup
    (1, 6, 7, 64, 32, 1, 1, 0, 1, False, False, 96, 0, "f16"),
    (2, 9, 8, 64, 64, 3, 1, 1, 1, True, False, 96, 0, "f16"),
    (1, 20, 20, 64, 160, 3, 2, 1, 1, False, False, 96, 0, "f16"),
    (3, 12, 12, 128, 264, 1, 1, 0, 0, False, False, 96, 2, "f16"),
    (2, 13, 11, 64, 256, 3, 1, 1, 1, True, False, 96, 1, "f16"),
    (2, 16, 16, 192, 320, 1, 1, 0, 1, False, False, 96, 3, "f16"),
    (3, 10, 10, 64, 384, 1, 1, 0, 1, True, False, 96, 1, "f16"),
    # stride-2 3x3 walks the taps in the class order of Y5ConvParams::tap_seq: two chunks per tap (C1 = 128), odd image sizes, several tiles per workgroup
    (3, 19, 21, 128, 96, 3, 2, 1, 1, False, False, 95, 1, "f16"),
    (3, 19, 21, 128, 96, 3, 2, 1, 1, False, False, 96, 1, "f16"),
    # C1 % 64 != 0 (the GEN loader: a K tile spans two taps): yolov5x's 80 / 160 and yolov5m's 96 channels; 3x3 s1 with residual, 3x3 s2, 1x1 (K tiles past the only tap)
    (2, 9, 10, 80, 80, 3, 1, 1, 1, True, False, 95, 1, "f16"),
    (2, 9, 10, 80, 80, 3, 1, 1, 1, True, False, 96, 1, "f16"),
    (2, 15, 13, 160, 96, 3, 2, 1, 1, False, False, 95, 0, "f16"),
    (2, 15, 13, 160, 96, 3, 2, 1, 1, False, False, 96, 0, "f16"),
    (3, 8, 9, 96, 136, 1, 1, 0, 1, False, False, 95, 1, "f16"),
    (3, 8, 9, 96, 136, 1, 1, 0, 0, False, False, 96, 1, "f16"),
    (1, 10, 10, 72, 64, 3, 1, 1, 1, False, False, 96, 0, "f16"),
] + [
    # every fp16 tile configuration on one shape with M, N tails and K = 9*64 (uniform) / 9*48 (table for BK64)
    (2, 9, 9, c1, 160, 3, 1, 1, 1, True, False, cfg, 2, "f16") for cfg in list(range(14)) + list(range(22, 30)) + list(range(35, 56)) for c1 in (64, 48)
]


def run_conv(lib, x_nchw, w, b, k, s, p, act, residual, up2, cfg, max_blocks, dt, ldx_extra=0, ldy_extra=0):
    B, C1, H, W = x_nchw.shape
    C2 = w.shape[0]
    npdt = np.float16 if dt == "f16" else np.float32
    tdt = torch.float16 if dt == "f16" else torch.float32
    ldx, ldy = C1 + ldx_extra, C2 + ldy_extra
    x = aligned((B, H, W, ldx), npdt, 7.0)
    x[..., :C1] = x_nchw.permute(0, 2, 3, 1).numpy().astype(npdt)
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, tdt)
    wp_a = aligned(wp.shape, npdt); wp_a[...] = wp.numpy()
    bp_a = aligned(bp.shape, np.float32); bp_a[...] = bp.numpy()
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = aligned((B, OH, OW, ldy), npdt, -3.0)
    res = None
    if residual:
        y[..., :C2] = detgen.uniform((B, OH, OW, C2), -1, 1, name="res").astype(npdt)
        res = y.copy()
    y2 = aligned((B, 2 * OH, 2 * OW, C2), npdt, -5.0) if up2 else None
    d = _lib.ConvDesc(dtype=_lib.Y5_F16 if dt == "f16" else _lib.Y5_F32, B=B, H=H, W=W, C1=C1, ldx=ldx, OH=OH, OW=OW,
                      C2=C2, ldy=ldy, KH=k, KW=k, SH=s, SW=s, PH=p, PW=p, act=act, Kpad=Kpad, Npad=Npad, ldr=ldy,
                      ld2=C2, cfg=cfg, max_blocks=max_blocks)
    rc = lib.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(wp_a), ptr(bp_a), ptr(y) if residual else None, ptr(y), ptr(y2), None)
    assert rc == 0, lib.y5_last_error()
    return y, y2, res


@pytest.mark.parametrize("case", CASES)
def test_conv_emulated_matches_torch(case):
    B, H, W, C1, C2, k, s, p, act, residual, up2, cfg, max_blocks, dt = case
    lib = emu()
    x = torch.from_numpy(detgen.uniform((B, C1, H, W), -1, 1, name="x"))
    w = torch.from_numpy(detgen.uniform((C2, C1, k, k), -0.3, 0.3, name="w"))
    b = torch.from_numpy(detgen.uniform((C2,), -0.5, 0.5, name="b"))
    if dt == "f16":
        x, w = x.half().float(), w.half().float()
    y, y2, res = run_conv(lib, x, w, b, k, s, p, act, residual, up2, cfg, max_blocks, dt, ldx_extra=8, ldy_extra=8)
    ref = F.conv2d(x, w, b, s, p)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1).numpy()
    if residual:
        ref = ref + res[..., :C2].astype(np.float32)
    tol = 3e-2 if dt == "f16" else 2e-5
    np.testing.assert_allclose(y[..., :C2].astype(np.float32), ref, rtol=tol, atol=tol)
    assert np.all(y[..., C2:] == (-3.0 if not residual else res[..., C2:]))  # neighbouring channels untouched
    if up2:
        up = np.repeat(np.repeat(y[..., :C2], 2, axis=1), 2, axis=2)
        assert np.array_equal(up, y2)


def test_conv_layer0_pair_view():
    """Layer 0 (k6 s2 p2, C=3): NHWC4 input viewed as (H, W/2, 8), filter as (6, 3, 8), stride (2,1), pad (2,1)."""
    lib = emu()
    B, H, W = 1, 16, 16
    x = torch.from_numpy(detgen.uniform((B, 3, H, W), 0, 1, name="x0")).half().float()
    w = torch.from_numpy(detgen.uniform((32, 3, 6, 6), -0.2, 0.2, name="w0")).half().float()
    b = torch.from_numpy(detgen.uniform((32,), -0.5, 0.5, name="b0"))
    xn = aligned((B, H, W, 4), np.float16, 0.0)
    xn[..., :3] = x.permute(0, 2, 3, 1).numpy().astype(np.float16)
    w4 = torch.zeros((32, 4, 6, 6)); w4[:, :3] = w
    # (C2, C, KH, KW) -> k order (kh, kw, c) with (kw, c) regrouped as (kw/2, 8)
    wp, bp, K, Kpad, Npad = pack_conv_weight(w4, b, torch.float16)
    assert K == 144 and Kpad == 192
    wp_a = aligned(wp.shape, np.float16); wp_a[...] = wp.numpy()
    bp_a = aligned(bp.shape, np.float32); bp_a[...] = bp.numpy()
    OH = OW = 8
    y = aligned((B, OH, OW, 32), np.float16)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W // 2, C1=8, ldx=8, OH=OH, OW=OW, C2=32, ldy=32, KH=6, KW=3, SH=2, SW=1,
                      PH=2, PW=1, act=1, Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=-1, max_blocks=0)
    rc = lib.y5_conv2d_fwd(C.byref(d), ptr(xn), ptr(wp_a), ptr(bp_a), None, ptr(y), None, None)
    assert rc == 0, lib.y5_last_error()
    ref = F.silu(F.conv2d(x, w, b, 2, 2)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y.astype(np.float32), ref, rtol=2e-2, atol=2e-2)


def test_conv_rejects_bad_args():
    lib = emu()
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=1, H=4, W=4, C1=12, ldx=12, OH=4, OW=4, C2=32, ldy=32, KH=1, KW=1, SH=1, SW=1,
                      PH=0, PW=0, act=1, Kpad=64, Npad=32, cfg=-1)
    a = aligned((64,), np.float16)
    assert lib.y5_conv2d_fwd(C.byref(d), ptr(a), ptr(a), ptr(a), None, ptr(a), None, None) == -1
    assert b"16 bytes" in lib.y5_last_error()


@pytest.mark.parametrize("B,H,W,C2,max_blocks", [(1, 8, 64, 32, 1), (2, 12, 128, 16, 2), (1, 6, 64, 48, 1), (3, 4, 64, 64, 1)])
def test_conv_stem_direct_nchw(B, H, W, C2, max_blocks):
    """y5_conv_stem_fwd (conv_stem.h): k6 s2 p2 straight from the NCHW batch, all four borders, Npad 32 and 64."""
    from yolov5_amd.packing import pack_stem_weight

    lib = emu()
    x = torch.from_numpy(detgen.uniform((B, 3, H, W), 0, 1, name="xs")).half()
    w = torch.from_numpy(detgen.uniform((C2, 3, 6, 6), -0.3, 0.3, name="ws")).half().float()
    b = torch.from_numpy(detgen.uniform((C2,), -0.5, 0.5, name="bs"))
    wp, bp, npad = pack_stem_weight(w, b)
    xa = aligned(x.shape, np.float16); xa[...] = x.numpy()
    wa = aligned(wp.shape, np.float16); wa[...] = wp.numpy()
    ba = aligned(bp.shape, np.float32); ba[...] = bp.numpy()
    ldy = C2 + 8
    y = aligned((B, H // 2, W // 2, ldy), np.float16, -3.0)
    rc = lib.y5_conv_stem_fwd(ptr(xa), B, H, W, ptr(wa), ptr(ba), C2, npad, ptr(y), ldy, max_blocks, None)
    assert rc == 0, lib.y5_last_error()
    ref = F.silu(F.conv2d(x.float(), w, b, 2, 2)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y[..., :C2].astype(np.float32), ref, rtol=2e-2, atol=2e-2)
    assert np.all(y[..., C2:] == -3.0)


@pytest.mark.parametrize("B,H,W,C2", [(1, 8, 64, 32), (2, 6, 128, 16), (1, 4, 64, 64)])
def test_conv_stem_raw_and_device_filter_pack(B, H, W, C2):
    """y5_conv_stem_fwd_raw (train mode: no bias, no activation) on a filter packed by the y5_filter_jobs kind-3 job (fp32 (C2, 3, 6, 6) master weights ->
    [Npad][144] fp16, the layout of packing.pack_stem_weight) against torch's conv2d."""
    from yolov5_amd.packing import pack_stem_weight, round_up

    lib = emu()
    x = torch.from_numpy(detgen.uniform((B, 3, H, W), 0, 1, name="xr")).half()
    w = torch.from_numpy(detgen.uniform((C2, 3, 6, 6), -0.3, 0.3, name="wr"))
    npad = round_up(C2, 32)
    w32 = aligned(w.shape, np.float32); w32[...] = w.numpy()
    wa = aligned((npad * 144,), np.float16, 7.0)
    job = (_lib.FilterJob * 1)()
    j = job[0]
    j.src, j.dst, j.total, j.kind, j.C2, j.C1, j.KH, j.KW, j.Kpad, j.Npad = w32.ctypes.data, wa.ctypes.data, npad * 144, 3, C2, 3, 6, 6, 144, npad
    tab = aligned((C.sizeof(job),), np.uint8); tab[...] = np.frombuffer(job, dtype=np.uint8)
    assert lib.y5_filter_jobs(ptr(tab), 1, npad * 144, None) == 0, lib.y5_last_error()
    ref_w, _, _ = pack_stem_weight(w, None)
    assert np.array_equal(wa.reshape(npad, 144), ref_w.numpy())
    xa = aligned(x.shape, np.float16); xa[...] = x.numpy()
    ldy = C2 + 8
    y = aligned((B, H // 2, W // 2, ldy), np.float16, -3.0)
    rc = lib.y5_conv_stem_fwd_raw(ptr(xa), B, H, W, ptr(wa), C2, npad, ptr(y), ldy, 0, None)
    assert rc == 0, lib.y5_last_error()
    ref = F.conv2d(x.float(), w.half().float(), None, 2, 2).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y[..., :C2].astype(np.float32), ref, rtol=5e-3, atol=5e-3)
    assert np.all(y[..., C2:] == -3.0)


@pytest.mark.parametrize("B,H,W,c_up,c_hi,C2,cfg,max_blocks", [
    (2, 8, 12, 64, 64, 128, 88, 0),     # producer / consumer ring, one chunk boundary at the source switch
    (1, 10, 6, 128, 64, 96, 88, 2),     # 4 low-resolution chunks + 2 plain ones, N tail, several tiles per workgroup
    (2, 6, 10, 64, 128, 160, 89, 0),    # 2-stage BK64 tile, two N tiles
    (3, 4, 4, 128, 128, 64, -1, 1),     # cfg -1 resolves to 89 for up_c > 0; one workgroup walks every tile
    # the 8-phase family's loader (conv_g8.h UP2, ids 95 / 96): source switch after one / two K tiles, N tails, several tiles per workgroup
    (2, 8, 12, 64, 64, 128, 95, 0),
    (2, 12, 12, 128, 128, 264, 95, 2),
    (1, 10, 6, 128, 64, 96, 96, 2),
    (2, 6, 10, 64, 128, 320, 96, 1),
])
def test_conv_virtual_upsample_concat(B, H, W, c_up, c_hi, C2, cfg, max_blocks):
    """Configurations 88 / 89 (conv_igemm.h UP2): the 1x1 convolution behind `nn.Upsample(2, 'nearest')` + `Concat` (models/yolov5s.yaml:36-38,41-43)
    reading the low-resolution tensor for input channels [0, c_up) -- against torch on the materialised concat.  The concat buffer's first c_up
    channels hold garbage (NaN) to prove they are never read."""
    lib = emu()
    lo = torch.from_numpy(detgen.uniform((B, c_up, H // 2, W // 2), -1, 1, name="lo")).half().float()
    hi = torch.from_numpy(detgen.uniform((B, c_hi, H, W), -1, 1, name="hi")).half().float()
    C1 = c_up + c_hi
    w = torch.from_numpy(detgen.uniform((C2, C1, 1, 1), -0.2, 0.2, name="wu")).half().float()
    b = torch.from_numpy(detgen.uniform((C2,), -0.5, 0.5, name="bu"))
    ld_lo, ldx, ldy = c_up + 16, C1 + 8, C2 + 8
    lo_a = aligned((B, H // 2, W // 2, ld_lo), np.float16, 9.0)
    lo_a[..., :c_up] = lo.permute(0, 2, 3, 1).numpy().astype(np.float16)
    x = aligned((B, H, W, ldx), np.float16, 7.0)
    x[..., :c_up] = np.nan
    x[..., c_up:C1] = hi.permute(0, 2, 3, 1).numpy().astype(np.float16)
    wp, bp, K, Kpad, Npad = pack_conv_weight(w, b, torch.float16)
    wp_a = aligned(wp.shape, np.float16); wp_a[...] = wp.numpy()
    bp_a = aligned(bp.shape, np.float32); bp_a[...] = bp.numpy()
    y = aligned((B, H, W, ldy), np.float16, -3.0)
    d = _lib.ConvDesc(dtype=_lib.Y5_F16, B=B, H=H, W=W, C1=C1, ldx=ldx, OH=H, OW=W, C2=C2, ldy=ldy, KH=1, KW=1, SH=1, SW=1, PH=0, PW=0, act=1,
                      Kpad=Kpad, Npad=Npad, ldr=0, ld2=0, cfg=cfg, max_blocks=max_blocks, up_c=c_up, ld_up=ld_lo)
    rc = lib.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(wp_a), ptr(bp_a), ptr(lo_a), ptr(y), None, None)
    assert rc == 0, lib.y5_last_error()
    cat = torch.cat((F.interpolate(lo, scale_factor=2, mode="nearest"), hi), 1)
    ref = F.silu(F.conv2d(cat, w, b)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y[..., :C2].astype(np.float32), ref, rtol=3e-2, atol=3e-2)
    assert np.all(y[..., C2:] == -3.0)
    # every other configuration refuses such a layer, and 88 / 89 refuse layers without one
    d.cfg = 43
    assert lib.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(wp_a), ptr(bp_a), ptr(lo_a), ptr(y), None, None) != 0
    d.cfg, d.up_c = 88, 0
    assert lib.y5_conv2d_fwd(C.byref(d), ptr(x), ptr(wp_a), ptr(bp_a), None, ptr(y), None, None) != 0


# ---- worst-case LDS-DMA landing model (tests/hipemu: Y5_EMU_ASYNC=1, latched per process -> child processes) -----------------------------------------
# In this mode an LDS-DMA load only lands when an `s_waitcnt vmcnt(N)` of its wave (or the end of the kernel) covers it: a counted wait that is one operation
# too lenient reads stale LDS.  EVERY convolution family is run through it, the four-wave streaming pointwise ids 14..21 / 56 included: their counted waits also
# count their global stores, which the kernels mark with Y5_EMU_VM_OP (csrc/y5_common.h) so that the model queues them as the hardware's counter does.
_ASYNC_CASES = list(range(len(CASES)))


def _run_cases(idx):
    for i in idx:
        test_conv_emulated_matches_torch(CASES[i])


@pytest.mark.parametrize("part", range(8))
def test_conv_worst_case_dma_landing(part):
    import os
    import subprocess
    import sys

    idx = _ASYNC_CASES[part::8]
    code = f"import tests.test_emu_conv as t; t._run_cases({idx})"
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, Y5_EMU_ASYNC="1"), capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
