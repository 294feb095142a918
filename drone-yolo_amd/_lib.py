"""ctypes binding of libdyolo.so (C-ABI declared in include/dyolo.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C drone-yolo_amd/csrc`` into
``drone-yolo_amd/lib/libdyolo.so``.  There is no fallback: if the library is missing, or a call
returns a non-zero status, this module raises — the product path never silently degrades to a
CPU or eager-PyTorch implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdyolo.so")

DY_BF16, DY_F16, DY_F32, DY_FP8, DY_F16X2 = 0, 1, 2, 3, 4
DY_ACT_NONE, DY_ACT_SILU, DY_ACT_SILU_L2E = 0, 1, 2
DY_MAX_LEVELS = 8
DY_WLAYOUT_ROWS, DY_WLAYOUT_HALO3X3, DY_WLAYOUT_FRAG1X1 = 0, 1, 2

_vp, _i32, _f32, _i64 = C.c_void_p, C.c_int32, C.c_float, C.c_int64


class PackJob(C.Structure):
    """dy_pack_job: one convolution of dy_pack_conv_weights_batched (the arguments of dy_pack_conv_weights)."""
    _fields_ = [("w", C.c_void_p), ("s_co", C.c_int64), ("s_ci", C.c_int64), ("s_r", C.c_int64), ("s_q", C.c_int64),
                ("cout", C.c_int32), ("cin", C.c_int32), ("ksize", C.c_int32), ("transpose_flip", C.c_int32), ("cin_logical", C.c_int32),
                ("w_layout", C.c_int32), ("dst", C.c_void_p), ("dst_elems", C.c_int64)]


class ConvDesc(C.Structure):
    """Mirror of ``dy_conv_desc`` (include/dyolo.h)."""

    _fields_ = [
        ("x", _vp), ("w", _vp), ("bias", _vp), ("residual", _vp), ("y", _vp),
        ("batch", _i32), ("h", _i32), ("w_in", _i32), ("cin", _i32), ("ld_x", _i32),
        ("ho", _i32), ("wo", _i32), ("cout", _i32), ("ld_y", _i32), ("ld_res", _i32),
        ("ksize", _i32), ("stride", _i32), ("pad", _i32),
        ("groups", _i32), ("act", _i32), ("dtype", _i32), ("out_f32", _i32),
        ("k_pad", _i32), ("cout_pad", _i32), ("up2x", _i32),
        ("x2", _vp), ("ld_x2", _i32), ("cin_split", _i32), ("w_layout", _i32),
        ("w_scale", _vp), ("act_scale", _f32), ("bn_stats", _vp), ("y_dtype1", _i32),
        ("bnb_z", _vp), ("bnb_ld_z", _i32), ("bnb_act", _i32), ("bnb_mean", _vp), ("bnb_rstd", _vp), ("bnb_gamma", _vp), ("bnb_beta", _vp),
    ]  # fmt: skip


class BranchDesc(C.Structure):
    """Mirror of ``dy_branch_desc``."""

    _fields_ = [
        ("x", _vp), ("w3", _vp), ("b3", _vp), ("w1", _vp), ("b1", _vp), ("out", _vp),
        ("batch", _i32), ("h", _i32), ("w", _i32), ("ld_x", _i32), ("c_in", _i32), ("c_mid", _i32), ("nc", _i32), ("reg_max", _i32),
        ("kind", _i32), ("dtype", _i32), ("anchors", _i32), ("anchor0", _i32),
        ("stride", _f32),
        ("nms_workspace", _vp), ("nms_workspace_bytes", _i64), ("conf_thres", _f32), ("classes_mask", _vp), ("act_l2e", _i32),
    ]  # fmt: skip


class DecodeDesc(C.Structure):
    """Mirror of ``dy_decode_desc``."""

    _fields_ = [
        ("level", _vp * DY_MAX_LEVELS),
        ("h", _i32 * DY_MAX_LEVELS), ("w", _i32 * DY_MAX_LEVELS), ("ld", _i32 * DY_MAX_LEVELS),
        ("stride", _f32 * DY_MAX_LEVELS),
        ("n_levels", _i32), ("batch", _i32), ("nc", _i32), ("reg_max", _i32),
        ("out", _vp),
        ("nms_workspace", _vp), ("nms_workspace_bytes", _i64), ("conf_thres", _f32), ("classes_mask", _vp),
    ]  # fmt: skip


class HeadDecodeDesc(C.Structure):
    """Mirror of ``dy_head_decode_desc``."""

    _fields_ = [
        ("x_box", _vp * DY_MAX_LEVELS), ("x_cls", _vp * DY_MAX_LEVELS),
        ("w_box", _vp * DY_MAX_LEVELS), ("w_cls", _vp * DY_MAX_LEVELS),
        ("b_box", _vp * DY_MAX_LEVELS), ("b_cls", _vp * DY_MAX_LEVELS),
        ("ld_box", _i32 * DY_MAX_LEVELS), ("ld_cls", _i32 * DY_MAX_LEVELS),
        ("h", _i32 * DY_MAX_LEVELS), ("w", _i32 * DY_MAX_LEVELS),
        ("stride", _f32 * DY_MAX_LEVELS),
        ("n_levels", _i32), ("batch", _i32), ("nc", _i32), ("reg_max", _i32),
        ("c_box", _i32), ("c_cls", _i32), ("dtype", _i32),
        ("out", _vp),
        ("nms_workspace", _vp), ("nms_workspace_bytes", _i64), ("conf_thres", _f32), ("classes_mask", _vp),
    ]  # fmt: skip


class NmsDesc(C.Structure):
    """Mirror of ``dy_nms_desc``."""

    _fields_ = [
        ("pred", _vp),
        ("batch", _i32), ("nc", _i32), ("n_extra", _i32), ("anchors", _i32),
        ("conf_thres", _f32), ("iou_thres", _f32),
        ("max_det", _i32), ("max_nms", _i32),
        ("max_wh", _f32), ("agnostic", _i32),
        ("classes_mask", _vp),
        ("out", _vp), ("out_count", _vp), ("out_index", _vp),
        ("workspace", _vp), ("workspace_bytes", _i64), ("prefiltered", _i32), ("multi_label", _i32),
    ]  # fmt: skip


class LossDesc(C.Structure):
    """Mirror of ``dy_loss_desc``."""

    _fields_ = [
        ("level", _vp * DY_MAX_LEVELS),
        ("h", _i32 * DY_MAX_LEVELS), ("w", _i32 * DY_MAX_LEVELS), ("ld", _i32 * DY_MAX_LEVELS),
        ("stride", _f32 * DY_MAX_LEVELS),
        ("n_levels", _i32), ("batch", _i32), ("nc", _i32), ("reg_max", _i32),
        ("gt", _vp), ("gmax", _i32), ("topk", _i32),
        ("alpha", _f32), ("beta", _f32), ("box_gain", _f32), ("cls_gain", _f32), ("dfl_gain", _f32),
        ("out", _vp), ("out_owner", _vp), ("workspace", _vp), ("workspace_bytes", _i64),
        ("grad_level", _vp * DY_MAX_LEVELS), ("ld_grad", _i32 * DY_MAX_LEVELS),
    ]  # fmt: skip


class C2fDesc(C.Structure):
    """Mirror of ``dy_c2f_desc``."""

    _fields_ = [
        ("x", _vp), ("x_lo", _vp), ("y", _vp), ("w_cv1", _vp), ("w_m_cv1", _vp), ("w_m_cv2", _vp), ("w_cv2", _vp), ("bias", _vp),
        ("batch", _i32), ("h", _i32), ("w", _i32), ("cin", _i32), ("cin_lo", _i32), ("hidden", _i32), ("cout", _i32),
        ("ld_x", _i32), ("ld_x_lo", _i32), ("ld_y", _i32), ("shortcut", _i32), ("dtype", _i32), ("act_l2e", _i32),
    ]  # fmt: skip


class Stem2Desc(C.Structure):
    """Mirror of ``dy_stem2_desc``."""

    _fields_ = [
        ("x", _vp), ("w0", _vp), ("b0", _vp), ("w1", _vp), ("b1", _vp), ("y", _vp),
        ("n", _i32), ("h", _i32), ("w", _i32), ("ld_y", _i32), ("act0", _i32), ("act1", _i32), ("dtype", _i32),
        ("w1_scale", _vp),
    ]  # fmt: skip


class BnDesc(C.Structure):
    """Mirror of ``dy_bn_desc``."""

    _fields_ = [
        ("z", _vp), ("y", _vp), ("addend", _vp), ("dy", _vp), ("dz", _vp),
        ("rows", _i64),
        ("c", _i32), ("ld_z", _i32), ("ld_y", _i32), ("ld_add", _i32), ("ld_dy", _i32), ("ld_dz", _i32), ("dtype", _i32), ("act", _i32),
        ("gamma", _vp), ("beta", _vp), ("mean", _vp), ("rstd", _vp), ("running_mean", _vp), ("running_var", _vp),
        ("eps", _f32), ("momentum", _f32),
        ("dgamma", _vp), ("dbeta", _vp), ("workspace", _vp), ("workspace_bytes", _i64), ("partial_slabs", _i32),
    ]  # fmt: skip


# name -> (restype, argtypes); every symbol include/dyolo.h declares must appear here
# (tests/test_cabi.py checks both directions).
SIGNATURES = {
    "dy_version": (_i32, []),
    "dy_last_error_string": (C.c_char_p, []),
    "dy_last_kernel_name": (C.c_char_p, []),
    "dy_dtype_size": (_i32, [_i32]),
    "dy_pack_conv_weights_table_bytes": (C.c_int64, [_i32]),
    "dy_pack_conv_weights_table": (_i32, [C.POINTER(PackJob), _i32, _i32, _vp, C.c_int64, C.POINTER(C.c_int32)]),
    "dy_pack_conv_weights_batched": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "dy_conv_k_pad": (_i32, [_i32, _i32, _i32]),
    "dy_conv_cout_pad": (_i32, [_i32]),
    "dy_conv2d_nhwc": (_i32, [C.POINTER(ConvDesc), _vp]),
    "dy_conv_stats_written": (_i32, []),
    "dy_c2f_fused_supported": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "dy_c2f_fused": (_i32, [C.POINTER(C2fDesc), _vp]),
    "dy_stem2_fused_supported": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "dy_stem2_fused": (_i32, [C.POINTER(Stem2Desc), _vp]),
    "dy_stem_conv3x3s2_nchw": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_stem_conv3x3s2_nchw_u8": (_i32, [_vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_nchw_f32_to_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_resize_bilinear_u8_nchw_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_scale_img_nchw_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "dy_nhwc_to_nchw_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_upsample2x_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_copy_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_sppf_maxpool3": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_detect_decode": (_i32, [C.POINTER(DecodeDesc), _vp]),
    "dy_detect_head_decode_supported": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "dy_detect_head_decode": (_i32, [C.POINTER(HeadDecodeDesc), _vp]),
    "dy_nms_workspace_bytes": (_i64, [_i32, _i32]),
    "dy_nms": (_i32, [C.POINTER(NmsDesc), _vp]),
    "dy_scale_boxes": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "dy_detection_loss_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "dy_conv2d_wgrad_nhwc": (_i32, [C.POINTER(ConvDesc), _vp, _i32, _vp, _vp]),
    "dy_conv2d_wgrad_nhwc_ws": (_i32, [C.POINTER(ConvDesc), _vp, _i32, _vp, _vp, C.c_int64, _vp]),
    "dy_conv2d_wgrad_workspace_bytes": (C.c_int64, [C.POINTER(ConvDesc), _i32]),
    "dy_quantize_fp8_nhwc": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _vp]),
    "dy_pack_conv_weights": (_i32, [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _i32, _vp]),
    "dy_detect_branch_fused_supported": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "dy_detect_branch_fused": (_i32, [C.POINTER(BranchDesc), _vp]),
    "dy_nms_reset_counts": (_i32, [_vp, _i32, _vp]),
    "dy_conv2d_grouped_bwd_nhwc": (_i32, [C.POINTER(ConvDesc), _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    "dy_colsum": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dy_nchw_u8_to_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "dy_letterbox_u8_to_nchw_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "dy_tiles_u8_to_nchw_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "dy_rows_to_pred": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "dy_upsample2x_bwd_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_maxpool_bwd_nhwc": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_add_nhwc": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_add_dilated2_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_head_grad_split": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "dy_sumsq_f32": (_i32, [_vp, _i64, _vp, _vp]),
    "dy_sgd_step": (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _i32, _vp, _f32, _vp, _vp]),
    "dy_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _f32, _vp, _vp]),
    "dy_amp_update": (_i32, [_vp, _vp, _f32, _f32, _i32, _vp]),
    "dy_ema_update": (_i32, [_vp, _vp, _i64, _f32, _vp]),
    "dy_grad_sink_flush": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "dy_bn_workspace_bytes": (_i64, [_i32]),
    "dy_bn_train_fwd": (_i32, [C.POINTER(BnDesc), _vp]),
    "dy_bn_train_bwd": (_i32, [C.POINTER(BnDesc), _vp]),
    "dy_silu_fwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "dy_silu_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dy_detection_loss": (_i32, [C.POINTER(LossDesc), _vp]),
}

_lib = None


class DyoloError(RuntimeError):
    """A libdyolo call returned a non-zero ``dy_status``."""


def lib() -> C.CDLL:
    """Load (once) and return the library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C drone-yolo_amd/csrc`. "
                "There is no CPU / eager fallback for this path."
            )
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "libdyolo") -> None:
    if rc != 0:
        msg = lib().dy_last_error_string()
        raise DyoloError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")
