"""CPU: libdyolo.so loads and exports exactly the symbols include/dyolo.h declares (no compute calls)."""
import ctypes
import os
import re

from tests._util import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "dyolo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dy_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    import drone_yolo_amd._lib as L

    names = _declared()
    assert len(names) >= 15
    handle = L.lib()
    for n in names:
        assert hasattr(handle, n), f"{n} declared in dyolo.h but not exported by libdyolo.so"
    assert sorted(L.SIGNATURES) == names, "ctypes SIGNATURES out of sync with include/dyolo.h"


def test_host_only_entry_points():
    import drone_yolo_amd._lib as L

    h = L.lib()
    assert h.dy_version() == (0 << 16) | 1
    assert [h.dy_dtype_size(i) for i in (0, 1, 2, 7)] == [2, 2, 4, 0]
    # bf16: 8 chunks x 8 elements per K-step
    assert h.dy_conv_k_pad(64, 3, L.DY_BF16) == 576 and h.dy_conv_k_pad(32, 3, L.DY_BF16) == 320
    assert h.dy_conv_k_pad(8, 3, L.DY_BF16) == 128 and h.dy_conv_k_pad(96, 1, L.DY_F32) == 96
    assert [h.dy_conv_cout_pad(c) for c in (10, 64, 65, 512)] == [64, 64, 128, 512]
    assert h.dy_nms_workspace_bytes(2, 34000) == 256 + 2 * 65536 * 8 + 136192
    assert h.dy_last_error_string() is not None


def test_descriptor_validation_without_gpu():
    """Argument checks run before any HIP call, so they are testable on CPU."""
    import drone_yolo_amd._lib as L

    h = L.lib()
    d = L.ConvDesc()
    assert h.dy_conv2d_nhwc(ctypes.byref(d), None) == -1  # DY_ERR_INVALID_ARG: null pointers
    assert b"null" in h.dy_last_error_string()
    n = L.NmsDesc()
    assert h.dy_nms(ctypes.byref(n), None) == -1
    assert h.dy_conv2d_nhwc(None, None) == -1


def test_struct_layout_matches_header():
    """sizeof() of the ctypes mirrors must equal the C structs (checked against a tiny C program)."""
    import subprocess
    import tempfile

    import drone_yolo_amd._lib as L

    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "s.c")
        open(src, "w").write('#include <stdio.h>\n#include "dyolo.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                             "sizeof(dy_conv_desc), sizeof(dy_decode_desc), sizeof(dy_nms_desc), sizeof(dy_loss_desc), "
                             "sizeof(dy_head_decode_desc), sizeof(dy_bn_desc), sizeof(dy_c2f_desc), sizeof(dy_stem2_desc));return 0;}\n")
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(L.ConvDesc), ctypes.sizeof(L.DecodeDesc), ctypes.sizeof(L.NmsDesc), ctypes.sizeof(L.LossDesc),
                     ctypes.sizeof(L.HeadDecodeDesc), ctypes.sizeof(L.BnDesc), ctypes.sizeof(L.C2fDesc), ctypes.sizeof(L.Stem2Desc)]


def test_new_entry_points_validate_without_gpu():
    """The training / fusion / preprocessing entry points refuse null or inconsistent descriptors before touching HIP."""
    import drone_yolo_amd._lib as L

    h = L.lib()
    null = None
    assert h.dy_c2f_fused(ctypes.byref(L.C2fDesc()), null) == -1 and b"null" in h.dy_last_error_string()
    assert h.dy_detect_head_decode(ctypes.byref(L.HeadDecodeDesc()), null) == -1
    assert h.dy_detection_loss(ctypes.byref(L.LossDesc()), null) == -1
    assert h.dy_bn_train_fwd(ctypes.byref(L.BnDesc()), null) == -1 and h.dy_bn_train_bwd(ctypes.byref(L.BnDesc()), null) == -1
    assert h.dy_conv2d_wgrad_nhwc(ctypes.byref(L.ConvDesc()), null, 0, null, null) == -1
    assert h.dy_conv2d_wgrad_nhwc_ws(ctypes.byref(L.ConvDesc()), null, 0, null, null, 0, null) == -1
    assert h.dy_conv2d_wgrad_workspace_bytes(ctypes.byref(L.ConvDesc()), 0) == -1
    d = L.ConvDesc()  # 64 -> 64 3x3 at 160x160, batch 64, bf16: 256 pixel slabs x (64 x 9 x 64) fp32 partial sums
    d.batch, d.h, d.w_in, d.cin, d.ld_x, d.ho, d.wo, d.cout, d.ksize, d.stride, d.pad, d.groups, d.dtype = 64, 160, 160, 64, 64, 160, 160, 64, 3, 1, 1, 1, L.DY_BF16
    assert h.dy_conv2d_wgrad_workspace_bytes(ctypes.byref(d), 64) == 256 * 64 * 9 * 64 * 4
    d.ksize, d.pad = 1, 0  # 1x1 stride 1: 512 slabs x 4 pixel splits inside a workgroup x (64 x 64)
    assert h.dy_conv2d_wgrad_workspace_bytes(ctypes.byref(d), 64) == 512 * 4 * 64 * 64 * 4
    s = L.ConvDesc()  # the image stem (8 <- 3 padded channels, 32 couts, 3x3 stride 2 at 640x640, batch 64): one partial per workgroup of conv_wgrad_stem_kernel
    s.batch, s.h, s.w_in, s.cin, s.ld_x, s.ho, s.wo, s.cout, s.ksize, s.stride, s.pad, s.groups, s.dtype = 64, 640, 640, 8, 8, 320, 320, 32, 3, 2, 1, 1, L.DY_BF16
    assert h.dy_conv2d_wgrad_workspace_bytes(ctypes.byref(s), 32) == 768 * 32 * 9 * 8 * 4
    d.dtype = L.DY_F32  # the per-tap kernel (fp32, strided, k x k layers) keeps its atomics
    assert h.dy_conv2d_wgrad_workspace_bytes(ctypes.byref(d), 64) == 0
    assert h.dy_letterbox_u8_to_nchw_f32(null, null, 1, 8, 8, 8, 8, 0, 0, 8, 8, 1, 114.0, null) == -1
    assert h.dy_sgd_step(null, null, null, 10, 0.1, 0.9, 0.0, 1, 1, null, 10.0, null, null) == -1
    assert h.dy_adamw_step(null, null, null, null, 10, 0.1, 0.9, 0.999, 1e-8, 0.0, 0, null, 10.0, null, null) == -1
    assert h.dy_amp_update(null, null, 2.0, 0.5, 2000, null) == -1 and b"dy_amp_update" in h.dy_last_error_string()
    assert h.dy_last_kernel_name() is not None
    assert h.dy_stem_conv3x3s2_nchw_u8(null, 255.0, null, null, null, 1, 3, 8, 8, 32, 32, 0, L.DY_BF16, null) == -1
    assert h.dy_add_dilated2_nhwc(null, null, 1, 4, 4, 8, 8, 64, 64, 64, L.DY_BF16, null) == -1
    assert h.dy_head_grad_split(null, 80, 10, 64, 10, 16, null, null, 64, null, 16, L.DY_BF16, null) == -1
    assert h.dy_pack_conv_weights_table_bytes(0) == -1 and h.dy_pack_conv_weights_table_bytes(3) % 3 == 0
    assert h.dy_pack_conv_weights_table(None, 1, L.DY_BF16, None, 0, None) == -1 and h.dy_pack_conv_weights_batched(None, 1, 1, L.DY_BF16, null) == -1
    # shape support queries are pure host functions
    assert h.dy_c2f_fused_supported(64, 0, 32, 64, 1, L.DY_BF16) == 1 and h.dy_c2f_fused_supported(64, 0, 32, 64, 2, L.DY_BF16) == 0
    assert h.dy_c2f_fused_supported(64, 0, 32, 64, 1, L.DY_F32) == 0 and h.dy_c2f_fused_supported(192, 0, 32, 64, 1, L.DY_BF16) == 0
    assert h.dy_c2f_fused_supported(192, 128, 32, 64, 1, L.DY_F16) == 1 and h.dy_c2f_fused_supported(192, 64, 32, 64, 1, L.DY_F16) == 0
    assert h.dy_detect_head_decode_supported(64, 64, 10, 16, L.DY_BF16) == 1 and h.dy_detect_head_decode_supported(64, 80, 80, 16, L.DY_BF16) == 0
    assert h.dy_stem2_fused(ctypes.byref(L.Stem2Desc()), null) == -1
    assert h.dy_stem2_fused_supported(3, 32, 64, 640, 640, L.DY_BF16) == 1 and h.dy_stem2_fused_supported(3, 32, 64, 642, 640, L.DY_BF16) == 0
    assert h.dy_stem2_fused_supported(3, 48, 96, 640, 640, L.DY_BF16) == 0 and h.dy_stem2_fused_supported(3, 32, 64, 640, 640, L.DY_F32) == 0
    assert h.dy_bn_workspace_bytes(64) == (1 + 1024) * 2 * 64 * 8 and h.dy_bn_workspace_bytes(0) == -1
    assert h.dy_detection_loss_workspace_bytes(2, 340, 7, 10) > 0 and h.dy_detection_loss_workspace_bytes(0, 340, 7, 10) == -1


def test_split_float16_type_validates_without_gpu():
    """DY_F16X2 (round 5): size / padding rules and the descriptor checks of the entry points built for it, before any HIP call."""
    import drone_yolo_amd._lib as L

    h = L.lib()
    assert L.DY_F16X2 == 4 and h.dy_dtype_size(L.DY_F16X2) == 4
    # a K-step is 32 channels (four hi / lo chunk pairs of a 128-byte row): 3x3 x 64 -> 576, 3x3 x 8 (the padded image) -> 96, 1x1 x 96 -> 96
    assert h.dy_conv_k_pad(64, 3, L.DY_F16X2) == 576 and h.dy_conv_k_pad(8, 3, L.DY_F16X2) == 96 and h.dy_conv_k_pad(96, 1, L.DY_F16X2) == 96
    assert h.dy_detect_head_decode_supported(64, 64, 10, 16, L.DY_F16X2) == 1 and h.dy_detect_head_decode_supported(64, 80, 10, 16, L.DY_F16X2) == 0
    assert h.dy_detect_head_decode_supported(128, 64, 10, 16, L.DY_F16X2) == 0
    assert h.dy_c2f_fused_supported(64, 0, 32, 64, 1, L.DY_F16X2) == 0  # the fused 16-bit C2f kernel: not for the type
    assert h.dy_stem2_fused_supported(3, 32, 64, 640, 640, L.DY_F16X2) == 1  # r05: the fused image-layer pair is (stem2_split_kernel) ...
    s2 = L.Stem2Desc()
    s2.x = s2.w0 = s2.b0 = s2.w1 = s2.b1 = s2.y = ctypes.cast((ctypes.c_float * 64)(), ctypes.c_void_p)
    s2.n, s2.h, s2.w, s2.ld_y, s2.act0, s2.act1, s2.dtype = 1, 64, 64, 64, 1, 1, L.DY_F16X2
    assert h.dy_stem2_fused(ctypes.byref(s2), None) == -1 and b"w1_scale" in h.dy_last_error_string()  # ... and needs layer 1's inverse row scales
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    d = L.ConvDesc()
    d.x = d.w = d.bias = d.y = p
    d.batch, d.h, d.w_in, d.cin, d.ld_x, d.ho, d.wo, d.cout, d.ld_y = 1, 8, 8, 64, 64, 8, 8, 64, 64
    d.ksize, d.stride, d.pad, d.groups, d.dtype, d.k_pad, d.cout_pad = 3, 1, 1, 1, L.DY_F16X2, 576, 64
    assert h.dy_conv2d_nhwc(ctypes.byref(d), None) == -1 and b"w_scale" in h.dy_last_error_string()  # the inverse row scales are part of the type
    d.groups = 2
    assert h.dy_conv2d_nhwc(ctypes.byref(d), None) == -2 and b"cout == groups" in h.dy_last_error_string()  # DY_ERR_UNSUPPORTED: of the grouped forms only DWConv's is built (r05)
    d.groups, d.cin, d.k_pad = 1, 12, 128
    assert h.dy_conv2d_nhwc(ctypes.byref(d), None) in (-1, -2)  # channels not in whole groups of 8
    assert h.dy_resize_bilinear_u8_nchw_f32(None, None, 1, 3, 8, 8, 16, 16, None) == -1
    assert h.dy_nchw_f32_to_nhwc(p, p, 1, 3, 8, 8, 4, 4, L.DY_F16X2, None) == -1  # c_pad must be whole groups of 8 channels
    assert h.dy_sppf_maxpool3(p, p, p, p, 1, 8, 8, 12, 12, 5, L.DY_F16X2, None) == -1
