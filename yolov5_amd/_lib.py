"""ctypes binding of libyolov5_hip.so (C-ABI declared in include/yolov5_hip.h).

The library is built in-tree by `make -C yolov5_amd/csrc` (or `__graft_entry__.build()`); it is NOT optional:
`lib()` raises RuntimeError when it cannot be loaded -- there is no eager/PyTorch fallback for any op.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
def _tokens(var):
    return {t.strip() for t in os.environ.get(var, "").split(",") if t.strip()}


def experimental(name):
    """Opt-in forms that were built, are parity-tested and LOST their A/B (or are measurement aids): `Y5_EXPERIMENTAL=<comma list>` -- streamk, cv3_128,
    head_branch, split2, h3_s2 (tuner candidates: the stride-2 halo ids), ddp_dry, stats_debug.  Read at call time."""
    return name in _tokens("Y5_EXPERIMENTAL")


def disabled(name):
    """Off-switches of shipped features, for A/B runs: `Y5_DISABLE=<comma list>` -- raw_view, train_stem, stem, fresh_outputs, obj_hint, virtual_up,
    head_deep, pipe_overlap, bn_fused_stats.  Read at call time."""
    return name in _tokens("Y5_DISABLE")


LIB_PATH = os.environ.get("Y5_LIB_PATH") or os.path.join(_HERE, "libyolov5_hip.so")  # override: kernel experiments only

Y5_F16, Y5_F32, Y5_U8 = 0, 1, 2
Y5_OK, Y5_ERR_BAD_ARG, Y5_ERR_UNSUPPORTED, Y5_ERR_RUNTIME, Y5_ERR_WORKSPACE = 0, -1, -2, -3, -4   # y5_status (include/yolov5_hip.h)
NMS_MULTI_LABEL, NMS_AGNOSTIC = 1, 2


class MaskImg(C.Structure):
    """y5_mask_img (include/yolov5_hip.h): one image of y5_process_mask_batch."""

    _fields_ = [("masks_in", C.c_void_p), ("boxes", C.c_void_p), ("ld_m", C.c_int), ("ld_b", C.c_int), ("n", C.c_int)]


class ConvDesc(C.Structure):
    """y5_conv_desc (include/yolov5_hip.h)."""

    _fields_ = [(n, C.c_int) for n in (
        "dtype", "B", "H", "W", "C1", "ldx", "OH", "OW", "C2", "ldy", "KH", "KW", "SH", "SW", "PH", "PW",
        "act", "Kpad", "Npad", "ldr", "ld2", "cfg", "max_blocks",
        "out_mul_h", "out_mul_w", "out_off_h", "out_off_w", "out_H", "out_W", "split_n", "up_c", "ld_up")]


class LossDesc(C.Structure):
    """y5_loss_desc (include/yolov5_hip.h)."""

    _fields_ = [("dtype", C.c_int), ("nl", C.c_int), ("na", C.c_int), ("nc", C.c_int), ("bs", C.c_int),
                ("ny", C.c_int * 5), ("nx", C.c_int * 5), ("anchors", C.c_float * 80), ("balance", C.c_float * 5),
                ("hyp_box", C.c_float), ("hyp_obj", C.c_float), ("hyp_cls", C.c_float), ("cls_pw", C.c_float),
                ("obj_pw", C.c_float), ("anchor_t", C.c_float), ("cp", C.c_float), ("cn", C.c_float), ("fl_gamma", C.c_float)]


class FilterJob(C.Structure):
    """include/yolov5_hip.h: y5_filter_job (one filter re-pack / weight-gradient unpack of a multi-filter launch)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("total", C.c_longlong)] + \
               [(n, C.c_int) for n in ("kind", "C2", "C1", "KH", "KW", "C1_view", "C2_view", "Kpad", "Npad", "nth", "ntw")] + \
               [("th", C.c_int * 8), ("tw", C.c_int * 8), ("reserved", C.c_int)]


class LetterboxJob(C.Structure):
    """include/yolov5_hip.h: y5_letterbox_job (one image of a y5_letterbox_batch launch)."""
    _fields_ = [("src", C.c_void_p)] + [(n, C.c_int) for n in ("h0", "w0", "stride", "nw", "nh", "top", "left")]


class MosaicJob(C.Structure):
    """include/yolov5_hip.h: y5_mosaic_job (one output image of a y5_mosaic_batch launch)."""
    _fields_ = [("src", C.c_void_p * 4)] + [(n, C.c_int * 4) for n in ("h0", "w0", "stride", "rh", "rw", "x1a", "y1a", "x2a", "y2a", "x1b", "y1b")] + \
               [("A", C.c_double * 6), ("lut", (C.c_ubyte * 256) * 3), ("hsv", C.c_int), ("flipud", C.c_int), ("fliplr", C.c_int), ("canvas", C.c_int),
                ("mix_r", C.c_double), ("mix_job", C.c_int), ("reserved", C.c_int)]


class MtTensor(C.Structure):
    """include/yolov5_hip.h: y5_mt_tensor (one row of the fused optimizer's device-resident tensor table)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("mom", C.c_void_p), ("ema", C.c_void_p), ("n", C.c_longlong),
                ("group", C.c_int), ("reserved", C.c_int)]


EXPORTS = {
    # name: (restype, argtypes)
    "y5_version": (C.c_int, []),
    "y5_last_error": (C.c_char_p, []),
    "y5_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "y5_conv2d_time": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "y5_conv_stem_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_int, C.c_void_p]),
    "y5_conv_stem_fwd_raw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y5_conv_num_cfgs": (C.c_int, []),
    "y5_probe_mfma": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]),
    "y5_set_cu_budget": (C.c_int, [C.c_int]),
    "y5_conv_sk_workspace_bytes": (C.c_size_t, []),
    "y5_conv_set_sk_workspace": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_conv_cfg_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "y5_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_float, C.c_void_p]),
    "y5_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p]),
    "y5_sppf_pool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_upsample2x": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p]),
    "y5_copy_slice": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_detect_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_longlong, C.c_longlong,
                                   C.c_void_p, C.c_void_p]),
    "y5_detect_decode_hint": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_longlong, C.c_longlong,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "y5_nms_batched_hint": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                      C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "y5_plan_set_obj_hint": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "y5_nms_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "y5_nms_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                 C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_nhwc_to_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_raw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_upsample2x_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_add_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_sppf_pool_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_train_glue_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y5_pack_dgrad_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int,
                                       C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y5_unpack_conv_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_memset_zero": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_filter_jobs": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]),
    "y5_mt_workspace_bytes": (C.c_size_t, [C.c_int, C.c_longlong]),
    "y5_mt_grad_norm": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_mt_sgd_step": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.c_float,
                                 C.c_void_p, C.c_float, C.c_void_p]),
    "y5_mt_lerp": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_void_p]),
    "y5_conv2d_wgrad": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "y5_conv2d_wgrad_det": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_conv2d_wgrad_ws_bytes": (C.c_longlong, [C.POINTER(ConvDesc), C.c_int]),
    "y5_bn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_longlong]),
    "y5_bn_silu_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    "y5_bn_silu_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_channel_sum": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_scale_img": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                               C.c_void_p, C.c_int, C.c_void_p]),
    "y5_tta_descale": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "y5_bn_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_sppf_cv1_pool_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    "y5_plan_add_sppf_cv1_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int]),
    "y5_conv2d_fwd_stats": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                     C.c_void_p]),
    "y5_bn_silu_fwd_from_partials": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_int, C.c_void_p]),
    "y5_bn_silu_fwd_from_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_int, C.c_void_p]),
    "y5_bn_bwd_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_bn_silu_bwd_from_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p]),
    "y5_loss_workspace_bytes": (C.c_size_t, [C.POINTER(LossDesc), C.c_int]),
    "y5_loss_obji_offset": (C.c_longlong, [C.POINTER(LossDesc), C.c_int]),
    "y5_loss_forward": (C.c_int, [C.POINTER(LossDesc), C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "y5_loss_backward": (C.c_int, [C.POINTER(LossDesc), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_void_p),
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "y5_loss_targets_layout": (C.c_int, [C.POINTER(LossDesc), C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_longlong)]),
    "y5_process_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "y5_process_mask_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(MaskImg), C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p]),
    "y5_mosaic_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y5_letterbox_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_val_match": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y5_scale_boxes_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "y5_plan_create": (C.c_void_p, []),
    "y5_plan_destroy": (None, [C.c_void_p]),
    "y5_plan_add_conv": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "y5_plan_add_conv_stem": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int]),
    "y5_detect_head_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float),
                                     C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p]),
    "y5_detect_head_fwd_hint": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float),
                                          C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]),
    "y5_plan_add_detect_head": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                          C.POINTER(C.c_float), C.c_void_p, C.c_longlong, C.c_longlong]),
    "y5_plan_add_nop": (C.c_int, [C.c_void_p]),
    "y5_plan_add_bottleneck": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "y5_conv_k3pw_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y5_conv_front_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_plan_add_conv_front": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "y5_plan_add_conv_k3pw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "y5_bottleneck_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_bottleneck_cv3_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "y5_plan_add_bottleneck_cv3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "y5_plan_set_input": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "y5_plan_set_branch": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "y5_plan_set_conv_cfg": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "y5_plan_set_anchors": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "y5_plan_add_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_float]),
    "y5_plan_add_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int]),
    "y5_plan_add_sppf_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int]),
    "y5_plan_add_upsample2x": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int]),
    "y5_plan_add_copy_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int]),
    "y5_plan_add_detect_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), C.c_void_p,
                                            C.c_int, C.c_longlong, C.c_longlong, C.c_void_p]),
    "y5_plan_size": (C.c_int, [C.c_void_p]),
    "y5_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "y5_plan_run_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y5_plan_capture": (C.c_int, [C.c_void_p, C.c_void_p]),
    "y5_plan_capture_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "y5_plan_launch_graph": (C.c_int, [C.c_void_p, C.c_void_p]),
    "y5_plan_time_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "y5_plan_profile_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "y5_plan_rebind_output": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y5_plan_select_graph": (C.c_int, [C.c_void_p, C.c_ulonglong]),
}


def bind(cdll):
    """Attach restype/argtypes for every symbol declared in include/yolov5_hip.h; fails loudly if one is missing."""
    for name, (res, args) in EXPORTS.items():
        fn = getattr(cdll, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return cdll


_lib = None
# TEST SEAM (tests/hipemu only): a host build of the SAME kernel sources (tests/hipemu/build.sh) bound with bind().  When set, the
# Python layer accepts CPU tensors and hands their host pointers to that library, so `-m "not gpu"` tests can drive the real
# wrappers / engines / loops end to end.  The product never sets it: without it every entry point refuses non-GPU tensors.
_test_lib = None
_test_backend = None  # engine backend factory that goes with it (tests.hipemu.backend.EmuBackend)


def use_test_library(handle, backend_factory=None):
    global _test_lib, _test_backend
    _test_lib, _test_backend = handle, backend_factory


def accepts(t) -> bool:
    """True when tensor `t` can be handed to the kernels: a GPU tensor (product), or any tensor under the test seam."""
    return bool(t.is_cuda) or (_test_lib is not None and t.device.type == "cpu")


def stream(device):
    """Current HIP stream of `device` as the void* the C-ABI takes (None for the host-emulated test library)."""
    import torch

    device = torch.device(device)
    if device.type != "cuda":
        if _test_lib is None:
            raise RuntimeError("yolov5_amd: tensors must live on the GPU (there is no CPU execution path)")
        return None
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def workspace(nbytes, device):
    """uint8 scratch tensor whose address is 256-byte aligned, as the C-ABI asks of every workspace (the GPU caching allocator
    aligns to 512 B already; host tensors of the test seam are 64-byte aligned, hence the slack)."""
    import torch

    t = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t[off:off + int(nbytes)]


def lib():
    """The loaded kernel library.  Raises RuntimeError (never falls back) if it is not built / not loadable."""
    global _lib
    if _test_lib is not None:
        return _test_lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"yolov5_amd: {LIB_PATH} not found -- build it with `make -C yolov5_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU/PyTorch fallback for the HIP kernels.")
        try:
            _lib = bind(C.CDLL(LIB_PATH))
        except OSError as e:  # e.g. libamdhip64 missing
            raise RuntimeError(f"yolov5_amd: cannot load {LIB_PATH}: {e}") from e
    return _lib


def check(rc: int, handle=None):
    """Translate a y5_status into a Python exception (message from y5_last_error())."""
    if rc != 0:
        h = handle or lib()
        msg = h.y5_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"yolov5_hip: {msg}")
        raise RuntimeError(f"yolov5_hip (status {rc}): {msg}")
