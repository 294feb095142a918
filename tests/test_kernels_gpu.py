"""GPU parity of every libdyolo kernel against a CPU fp32 restatement on the same (dtype-rounded) inputs.

Tolerances: the device accumulates in fp32 and rounds ONCE to the storage dtype, so against an fp32
CPU result on identically rounded inputs the error budget is one output rounding (2^-8 rel. for
bf16, 2^-11 for fp16) plus accumulation-order noise; fp32 storage must agree to ~1e-5 relative.
Integer / index outputs (NMS kept indices, classes, counts) are compared exactly.
"""
import zlib
import ast

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from drone_yolo_amd import hip_ops as H
from oracle import drone_yolo_oracle as O
from tests._util import golden, quantize, split_rows

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
RTOL = {torch.float32: 2e-5, torch.bfloat16: 6e-3, torch.float16: 1.2e-3}


def nhwc(t, dtype, dev, ld=None, c_off=0):
    """CPU NCHW fp32 -> device NHWC view (optionally a channel slice of a wider buffer)."""
    n, c, h, w = t.shape
    ld = ld or c
    buf = torch.zeros((n, h, w, ld), dtype=dtype, device=dev)
    buf[..., c_off : c_off + c] = t.permute(0, 2, 3, 1).to(dtype).to(dev)
    return buf.permute(0, 3, 1, 2)[:, c_off : c_off + c]


def back(t):
    return t.float().cpu().contiguous()


def check_close(got, ref, dtype, what, extra=1.0):
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    assert err <= RTOL[dtype] * extra * scale, f"{what}: max|err| {err:.4e} vs scale {scale:.3f} (tol {RTOL[dtype] * extra * scale:.4e})"


CONV_CASES = [
    # cin, cout, k, s, B, H, W, act, tag
    (8, 32, 3, 2, 2, 64, 64, True, "stem-like Cin=8"),
    (32, 64, 3, 2, 2, 40, 40, True, "repvgg-like s2"),
    (64, 64, 3, 1, 2, 40, 40, True, "3x3 s1 BN=64"),
    (32, 32, 3, 1, 1, 24, 20, True, "3x3 s1 BN=32"),
    (96, 64, 1, 1, 2, 40, 40, True, "1x1 K=96 (K tail)"),
    (768, 512, 1, 1, 2, 20, 20, True, "1x1 wide"),
    (256, 256, 3, 1, 1, 20, 20, True, "3x3 deep small-M"),
    (64, 64, 3, 1, 1, 13, 17, True, "odd spatial (M tail)"),
    (64, 64, 3, 1, 4, 160, 160, True, "large M (BM=128 path)"),
    (32, 32, 3, 1, 8, 160, 160, True, "large M BN=32"),
    (64, 16, 3, 1, 8, 160, 160, False, "large M BN=16 no act"),
    (16, 24, 3, 1, 1, 12, 12, True, "n-scale odd cout"),
    (24, 48, 1, 1, 1, 12, 12, True, "cin=24"),
    (48, 64, 3, 1, 2, 33, 37, True, "halo cin=48 odd dims"),
    (128, 128, 3, 2, 2, 40, 40, True, "halo s2 deep"),
    (64, 32, 3, 2, 1, 80, 80, True, "halo s2 BN=32"),
    (64, 64, 3, 1, 40, 40, 40, True, "halo many tiles per block"),
    (64, 64, 1, 1, 8, 160, 160, True, "1x1 stream 64->64 large M"),
    (192, 128, 1, 1, 3, 80, 80, True, "1x1 stream 192->128 (NF=8)"),
    (384, 128, 1, 1, 2, 40, 40, True, "1x1 stream 384->128 (12 k-groups)"),
    (256, 256, 1, 1, 2, 40, 40, True, "1x1 stream 256->256 (2 n-tiles)"),
    (128, 128, 1, 1, 1, 13, 11, True, "1x1 stream odd M tail"),
    (64, 8, 1, 1, 1, 20, 20, False, "1x1 stream cout=8 (NF=1)"),
    (64, 64, 3, 2, 2, 40, 36, True, "hreg s2 64->64"),
    (64, 128, 3, 2, 3, 33, 31, True, "hreg s2 64->128 odd dims"),
    (64, 64, 3, 2, 1, 8, 8, False, "hreg s2 tiny no act"),
    (64, 64, 3, 2, 2, 160, 160, True, "hreg s2 many tiles"),
    (128, 128, 3, 1, 2, 40, 40, True, "hreg 128->128 (four chunks in registers)"),
    (128, 64, 3, 1, 3, 33, 31, True, "hreg 128->64 odd dims"),
    (128, 256, 3, 1, 1, 8, 16, False, "hreg 128->256 tiny no act"),
    (128, 128, 3, 1, 24, 80, 80, True, "hreg 128->128 many tiles per block"),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[-1] for c in CONV_CASES])
def test_conv_matches_cpu(case, dtype, device):
    cin, cout, k, s, b, h, w, act, tag = case
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % 1000)
    x = quantize(torch.randn(b, cin, h, w, generator=g), dtype)
    wt = quantize(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5, dtype)
    bias = torch.randn(cout, generator=g) * 0.2
    ref = F.conv2d(x, wt, bias, s, k // 2)
    if act:
        ref = F.silu(ref)
    pc = H.PackedConv(wt, bias, s, k // 2, 1, act, dtype, device)
    y = H.conv2d(nhwc(x, dtype, device), pc)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(ref.shape)
    check_close(back(y), ref, dtype, f"conv {tag}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("how", ["residual", "out_f32", "unaligned_out", "x2", "up2x"])
def test_halo_layout_pack_falls_back_to_rows_for_calls_its_kernels_refuse(how, dtype, device):
    """A 16-bit 3x3 stride-2 layer with 64 input channels is packed in DY_WLAYOUT_HALO3X3 and has ONE kernel behind that layout
    (conv3x3_hreg_s2: no residual, no fp32 output, 16-byte output rows, one plain source).  A call outside that must run on the
    pack's DY_WLAYOUT_ROWS twin (PackedConv.for_call), not fail: every such call against the fp32 CPU convolution."""
    g = torch.Generator().manual_seed(zlib.crc32(how.encode()) % 1000)
    b, cin, cout, h, w = 2, 64, 128, 24, 20
    wt = quantize(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5, dtype)
    bias = torch.randn(cout, generator=g) * 0.2
    pc = H.PackedConv(wt, bias, 2, 1, 1, True, dtype, device)
    assert pc.layout == H._lib.DY_WLAYOUT_HALO3X3
    kw, res = {}, None
    if how == "x2":
        xa, xb = quantize(torch.randn(b, 32, h, w, generator=g), dtype), quantize(torch.randn(b, 32, h, w, generator=g), dtype)
        x, xin, kw = torch.cat([xa, xb], 1), nhwc(xa, dtype, device), {"x2": nhwc(xb, dtype, device)}
    elif how == "up2x":
        xl = quantize(torch.randn(b, cin, h // 2, w // 2, generator=g), dtype)
        x, xin, kw = F.interpolate(xl, scale_factor=2, mode="nearest"), nhwc(xl, dtype, device), {"up2x": True}
    else:
        x = quantize(torch.randn(b, cin, h, w, generator=g), dtype)
        xin = nhwc(x, dtype, device)
    ref = F.silu(F.conv2d(x, wt, bias, 2, 1))
    if how == "residual":
        r = quantize(torch.randn(ref.shape, generator=g), dtype)
        ref, kw = ref + r, {"residual": nhwc(r, dtype, device)}
    elif how == "out_f32":
        kw = {"out_f32": True}
    elif how == "unaligned_out":  # a channel slice whose pitch is not a multiple of eight elements: no 16-byte row stores
        kw = {"out": nhwc(torch.zeros(ref.shape), dtype, device, ld=cout + 4)}
    y = H.conv2d(xin, pc, **kw)
    torch.cuda.synchronize()
    assert pc._rows_pack is not None and pc._rows_pack.layout == H._lib.DY_WLAYOUT_ROWS
    check_close(back(y), ref, dtype, f"hreg_s2 fallback {how}", extra=2.0 if how == "residual" else 1.0)
    y2 = H.conv2d(nhwc(quantize(torch.randn(b, cin, h, w, generator=g), dtype), dtype, device), pc)  # the plain call still takes the fast kernel
    torch.cuda.synchronize()
    assert H.last_kernel_name().startswith("conv3x3_hreg_s2") and tuple(y2.shape) == (b, cout, h // 2, w // 2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("how", ["plain", "residual", "out_f32", "unaligned_out"])
def test_128_channel_3x3_runs_on_the_register_weight_kernel_or_its_rows_twin(how, dtype, device):
    """r04: a 16-bit 3x3 stride-1 layer with 128 input channels is packed for conv3x3_hreg's four-chunk form (DY_WLAYOUT_HALO3X3).  The
    plain call runs there; what that kernel declines (residual, fp32 output, an output pitch without 16-byte rows) runs on the pack's
    DY_WLAYOUT_ROWS twin (the virtual-flat GEMM / LDS-DMA GEMM).  Every call against the fp32 CPU convolution."""
    g = torch.Generator().manual_seed(zlib.crc32(("h128" + how).encode()) % 1000)
    b, cin, cout, h, w = 2, 128, 128, 24, 20
    wt = quantize(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5, dtype)
    bias = torch.randn(cout, generator=g) * 0.2
    # r05: bf16 packs WITH an activation (inference) stay on the virtual-flat GEMM (hip_ops.PackedConv: the four-chunk summation order cost the
    # s640 fixture's worst bf16 box its 0.998 IoU floor); the raw convolutions of a training step (no activation) and float16 keep the kernel
    act = dtype == torch.float16
    pc = H.PackedConv(wt, bias, 1, 1, 1, act, dtype, device)
    assert pc.layout == H._lib.DY_WLAYOUT_HALO3X3
    if dtype == torch.bfloat16:
        assert H.PackedConv(wt, bias, 1, 1, 1, True, dtype, device).layout == H._lib.DY_WLAYOUT_ROWS
    x = quantize(torch.randn(b, cin, h, w, generator=g), dtype)
    ref = F.conv2d(x, wt, bias, 1, 1)
    ref = F.silu(ref) if act else ref
    kw = {}
    if how == "residual":
        r = quantize(torch.randn(ref.shape, generator=g), dtype)
        ref, kw = ref + r, {"residual": nhwc(r, dtype, device)}
    elif how == "out_f32":
        kw = {"out_f32": True}
    elif how == "unaligned_out":
        kw = {"out": nhwc(torch.zeros(ref.shape), dtype, device, ld=cout + 4)}
    y = H.conv2d(nhwc(x, dtype, device), pc, **kw)
    torch.cuda.synchronize()
    name = H.last_kernel_name()
    assert name.startswith("conv3x3_hreg") if how == "plain" else not name.startswith("conv3x3_h"), name
    check_close(back(y), ref, dtype, f"hreg128 {how}", extra=2.0 if how == "residual" else 1.0)


FK_CASES = [
    # cin, cout, k, s, B, H, W, residual, tag — channel counts that are not whole 64-channel K-steps / 64-cout tiles (the n / m / x scales)
    (80, 80, 3, 1, 2, 24, 20, False, "x-scale P2 bottleneck 80->80 (tile 128x80)"),
    (160, 160, 3, 1, 2, 24, 20, False, "160->160 (tile 128x160)"),
    (160, 160, 3, 1, 1, 13, 17, True, "160->160 + residual, M tail"),
    (160, 320, 3, 2, 2, 25, 23, False, "RepVGG-like s2 160->320 odd dims"),
    (400, 160, 1, 1, 2, 24, 20, False, "C2f.cv2 400->160 (K tail inside a step)"),
    (160, 64, 3, 1, 1, 13, 17, False, "Detect box 160->64 (tile 128x64)"),
    (48, 96, 3, 1, 2, 16, 16, False, "m-scale 48->96 (two wrap-arounds per step)"),
    (240, 200, 1, 1, 1, 9, 7, False, "cout 200: masked last tile"),
    (160, 320, 3, 1, 1, 12, 12, False, "160->320 (two cout tiles)"),
    (96, 96, 1, 2, 2, 16, 16, False, "1x1 stride 2"),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", FK_CASES, ids=[c[-1] for c in FK_CASES])
def test_conv_flat_k_kernel_matches_cpu(case, dtype, device):
    """csrc/conv_gemm_fk.hip: K walked flat over (r, q, c) in 128-byte steps whatever taps they cross, cout tiles of 64 / 80 / 128 / 160
    masked on the last tile — against the fp32 CPU convolution on the same rounded operands (DY_WLAYOUT_ROWS forced: the layers of the
    n / m / x scales that used to fall to the generic kernel)."""
    cin, cout, k, s, b, h, w, res, tag = case
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % 1000)
    x = quantize(torch.randn(b, cin, h, w, generator=g), dtype)
    wt = quantize(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5, dtype)
    bias = torch.randn(cout, generator=g) * 0.2
    ref = F.silu(F.conv2d(x, wt, bias, s, k // 2))
    kw = {}
    if res:
        r = quantize(torch.randn(ref.shape, generator=g), dtype)
        ref, kw = ref + r, {"residual": nhwc(r, dtype, device)}
    pc = H.PackedConv(wt, bias, s, k // 2, 1, True, dtype, device, halo=False)
    y = H.conv2d(nhwc(x, dtype, device, ld=cin + 16, c_off=8), pc, **kw)  # a channel slice of a wider buffer
    torch.cuda.synchronize()
    assert H.last_kernel_name().startswith("conv_gemm_fk_kernel"), H.last_kernel_name()
    check_close(back(y), ref, dtype, f"flat-K {tag}", extra=2.0 if res else 1.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_conv_flat_k_two_sources_through_upsample(dtype, device):
    """Upsample + Concat folded into a 1x1 gather (C2f.cv1 of the neck) on the flat-K kernel: channels [0, 128) from a half-resolution
    map through the fused 2x nearest upsample, [128, 208) from the skip map; the split is a whole K-step, the total is not."""
    g = torch.Generator().manual_seed(11)
    b, c_lo, c_sk, cout, h, w = 2, 128, 80, 160, 12, 10
    lo = quantize(torch.randn(b, c_lo, h // 2, w // 2, generator=g), dtype)
    sk = quantize(torch.randn(b, c_sk, h, w, generator=g), dtype)
    wt = quantize(torch.randn(cout, c_lo + c_sk, 1, 1, generator=g) * (2.0 / (c_lo + c_sk)) ** 0.5, dtype)
    bias = torch.randn(cout, generator=g) * 0.2
    ref = F.silu(F.conv2d(torch.cat([F.interpolate(lo, scale_factor=2, mode="nearest"), sk], 1), wt, bias))
    pc = H.PackedConv(wt, bias, 1, 0, 1, True, dtype, device, halo=False)
    y = H.conv2d(nhwc(lo, dtype, device), pc, x2=nhwc(sk, dtype, device), up2x=True)
    torch.cuda.synchronize()
    assert H.last_kernel_name().startswith("conv_gemm_fk_kernel"), H.last_kernel_name()
    check_close(back(y), ref, dtype, "flat-K two sources")


GLDS_CASES = [
    # cin, cout, k, s, B, H, W, tag — ROWS layout, M >= 8192 and whole 128-byte K-steps: the LDS-DMA big-tile kernel
    (128, 128, 3, 1, 6, 40, 40, "3x3 s1 BN=128"),
    (256, 64, 3, 1, 6, 40, 40, "3x3 s1 BN=64"),
    (128, 256, 3, 2, 3, 111, 97, "3x3 s2 odd dims, M tail"),
    (768, 512, 1, 1, 24, 20, 20, "1x1 wide K"),
    (64, 192, 1, 1, 3, 64, 63, "1x1 cout=192 (BN=64), M tail"),
    (256, 256, 3, 1, 22, 20, 20, "3x3 on 20x20 maps"),
    # deep 3x3 stride-1: the virtual-flat-index kernel (conv3x3_vgemm.hip) — one tile, odd maps, the widest map it
    # takes (94), a map too wide for it (falls through to the per-tap gather), and more tiles than workgroups
    (128, 128, 3, 1, 3, 5, 7, "vgemm single tile 5x7"),
    (192, 128, 3, 1, 2, 33, 47, "vgemm cin=192 odd map"),
    (128, 256, 3, 1, 1, 9, 94, "vgemm widest map"),
    (128, 128, 3, 1, 1, 9, 96, "3x3 s1 too wide for vgemm"),
    (128, 128, 3, 1, 80, 40, 40, "vgemm persistent, 526 tiles"),
]


L2E_CASES = [(64, 64, 3, 1, 2, 40, 36, "hreg / halo 3x3"), (32, 64, 3, 2, 2, 40, 40, "halo s2"), (256, 256, 3, 1, 2, 20, 20, "vgemm"), (128, 128, 3, 1, 2, 20, 20, "hreg 128"), (256, 256, 1, 1, 2, 40, 40, "glds 1x1"),
             (192, 128, 1, 1, 2, 40, 40, "stream 1x1"), (16, 24, 3, 1, 1, 12, 12, "generic"), (128, 256, 3, 2, 2, 20, 20, "glds s2")]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", L2E_CASES, ids=[c[-1] for c in L2E_CASES])
def test_conv_silu_in_the_log2e_scaled_domain(case, dtype, device):
    """DY_ACT_SILU_L2E (include/dyolo.h): the accumulator is t = log2(e) * z and the epilogue returns t / (1 + 2^-t) = log2(e) * silu(z).
    Against the CPU: conv on the same rounded operands, t * sigmoid(t / log2 e) — every conv kernel's epilogue (the shapes pick them)."""
    cin, cout, k, s, b, h, w, tag = case
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % 1013)
    x = quantize(torch.randn(b, cin, h, w, generator=g), dtype)
    wt = quantize(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5, dtype)
    bias = torch.randn(cout, generator=g) * 0.2
    t = F.conv2d(x.double(), wt.double(), bias.double(), s, k // 2).float()
    ref = t * torch.sigmoid(t / H.LOG2E)
    pc = H.PackedConv(wt, bias, s, k // 2, 1, H.DY_ACT_SILU_L2E, dtype, device)
    got = H.conv2d(nhwc(x, dtype, device), pc)
    torch.cuda.synchronize()
    check_close(back(got), ref, dtype, f"silu_l2e {tag}", extra=2.0)


def test_scaled_activation_domain_folds(device):
    """hip_ops.domain_fold: biases (and the weights of the layer that reads raw data) times log2 e, SiLU -> DY_ACT_SILU_L2E, the Detect
    tails divided by log2 e; a two-layer chain Conv(raw image) -> Conv -> plain 1x1 in the scaled domain returns the true-unit result."""
    g = torch.Generator().manual_seed(4)
    dtype = torch.float16
    x = quantize(torch.rand(2, 8, 24, 20, generator=g), dtype)
    w0, b0 = torch.randn(32, 8, 3, 3, generator=g) * 0.2, torch.randn(32, generator=g) * 0.1
    w1, b1 = torch.randn(32, 32, 3, 3, generator=g) * 0.08, torch.randn(32, generator=g) * 0.1
    w2, b2 = torch.randn(16, 32, 1, 1, generator=g) * 0.2, torch.randn(16, generator=g) * 0.1
    ref = F.conv2d(F.silu(F.conv2d(F.silu(F.conv2d(x, w0, b0, 1, 1)), w1, b1, 1, 1)), w2, b2)
    assert H.domain_fold(w0, b0, True)[2] == H.DY_ACT_SILU  # off by default: modules on their own stay in the reference's units
    with H.scaled_activations(True):
        f0, f1, f2 = H.domain_fold(w0, b0, True, raw_input=True), H.domain_fold(w1, b1, True), H.domain_fold(w2, b2, False, raw_output=True)
        assert f0[2] == f1[2] == H.DY_ACT_SILU_L2E and f2[2] == H.DY_ACT_NONE
        assert torch.allclose(f0[0], w0 * H.LOG2E) and torch.allclose(f1[0], w1) and torch.allclose(f1[1], b1 * H.LOG2E) and torch.allclose(f2[0], w2 / H.LOG2E)
        pcs = [H.PackedConv(f[0], f[1], 1, k // 2, 1, f[2], dtype, device) for f, k in ((f0, 3), (f1, 3), (f2, 1))]
    y = H.conv2d(H.conv2d(H.conv2d(nhwc(x, dtype, device), pcs[0]), pcs[1]), pcs[2])
    torch.cuda.synchronize()
    check_close(back(y), ref, dtype, "scaled-domain chain", extra=6.0)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", GLDS_CASES, ids=[c[-1] for c in GLDS_CASES])
def test_conv_glds_big_tile_matches_cpu(case, dtype, device):
    cin, cout, k, s, b, h, w, tag = case
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % 1000)
    x = quantize(torch.randn(b, cin, h, w, generator=g), dtype)
    wt = quantize(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5, dtype)
    bias = torch.randn(cout, generator=g) * 0.2
    ref = F.silu(F.conv2d(x, wt, bias, s, k // 2))
    pc = H.PackedConv(wt, bias, s, k // 2, 1, True, dtype, device, halo=False)
    xd = nhwc(x, dtype, device, ld=cin + 16, c_off=8)  # a channel slice of a wider buffer
    y = H.conv2d(xd, pc)
    torch.cuda.synchronize()
    check_close(back(y), ref, dtype, f"glds conv {tag}")
    if k == 3 and s == 1 and cin == cout:  # Bottleneck residual
        y2 = H.conv2d(xd, pc, residual=xd)
        torch.cuda.synchronize()
        check_close(back(y2), ref + x, dtype, f"glds conv + residual {tag}")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_conv_glds_dual_source_upsample(dtype, device):
    """C2f.cv1 on cat(upsample2x(a), b) at a size that takes the LDS-DMA kernel (split on a K-step boundary)."""
    g = torch.Generator().manual_seed(19)
    a = quantize(torch.randn(3, 128, 30, 32, generator=g), dtype)
    bsk = quantize(torch.randn(3, 64, 60, 64, generator=g), dtype)
    wt = quantize(torch.randn(128, 192, 1, 1, generator=g) * 0.1, dtype)
    bias = torch.randn(128, generator=g) * 0.1
    pc = H.PackedConv(wt, bias, 1, 0, 1, True, dtype, device, halo=False)
    y = H.conv2d(nhwc(a, dtype, device), pc, up2x=True, x2=nhwc(bsk, dtype, device))
    torch.cuda.synchronize()
    ref = F.silu(F.conv2d(torch.cat((F.interpolate(a, scale_factor=2.0, mode="nearest"), bsk), 1), wt, bias))
    check_close(back(y), ref, dtype, "glds dual-source up2x conv")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_conv_slices_residual_and_f32_out(dtype, device):
    g = torch.Generator().manual_seed(7)
    b, c, h, w = 2, 32, 20, 24
    x = quantize(torch.randn(b, c, h, w, generator=g), dtype)
    wt = quantize(torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5, dtype)
    bias = torch.randn(c, generator=g) * 0.1
    pc = H.PackedConv(wt, bias, 1, 1, 1, True, dtype, device)
    # input = channel slice [32:64] of a 96-wide buffer, output = slice [64:96] of the same buffer, residual = input
    buf = torch.zeros((b, h, w, 96), dtype=dtype, device=device).permute(0, 3, 1, 2)
    buf[:, 32:64] = x.to(dtype).to(device)
    H.conv2d(buf[:, 32:64], pc, out=buf[:, 64:96], residual=buf[:, 32:64])
    torch.cuda.synchronize()
    ref = x + F.silu(F.conv2d(x, wt, bias, 1, 1))
    check_close(back(buf[:, 64:96]), ref, dtype, "conv slice+residual")
    assert float(buf[:, :32].float().abs().max()) == 0.0, "conv wrote outside its output slice"
    assert torch.equal(back(buf[:, 32:64]), x), "conv clobbered its input slice"
    # fp32 logits into an unaligned-width (nc=10) slice of a pitch-76 buffer (the Detect head layout)
    w10 = quantize(torch.randn(10, c, 1, 1, generator=g) * 0.2, dtype)
    b10 = torch.randn(10, generator=g)
    pc10 = H.PackedConv(w10, b10, 1, 0, 1, False, dtype, device)
    head = torch.full((b, h, w, 76), 7.0, dtype=torch.float32, device=device).permute(0, 3, 1, 2)
    H.conv2d(buf[:, 32:64], pc10, out=head[:, 64:74], out_f32=True)
    torch.cuda.synchronize()
    check_close(back(head[:, 64:74]), F.conv2d(x, w10, b10), torch.float32 if dtype == torch.float32 else dtype, "conv f32-out nc=10", extra=1.0)
    assert float((head[:, :64] - 7.0).abs().max()) == 0.0 and float((head[:, 74:] - 7.0).abs().max()) == 0.0
    # the Detect head shapes proper (cin 64): box bins 64 -> 64 and class logits 64 -> nc, fp32 out (streaming 1x1 kernel)
    x64 = quantize(torch.randn(b, 64, h, w, generator=g), dtype)
    xb = nhwc(x64, dtype, device)
    for cout, lo in ((64, 0), (10, 64)):
        wh = quantize(torch.randn(cout, 64, 1, 1, generator=g) * 0.2, dtype)
        bh = torch.randn(cout, generator=g)
        pch = H.PackedConv(wh, bh, 1, 0, 1, False, dtype, device, for_out_f32=True)
        head = torch.full((b, h, w, 76), 7.0, dtype=torch.float32, device=device).permute(0, 3, 1, 2)
        H.conv2d(xb, pch, out=head[:, lo : lo + cout], out_f32=True)
        torch.cuda.synchronize()
        check_close(back(head[:, lo : lo + cout]), F.conv2d(x64, wh, bh), dtype, f"head 64->{cout} f32-out")
        rest = torch.cat((head[:, :lo], head[:, lo + cout :]), 1)
        assert float((rest - 7.0).abs().max()) == 0.0, "head conv wrote outside its slice"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_conv_dual_source_upsample_gather(dtype, device):
    """cv1(cat(upsample2x(a), b)) with the Upsample and the Concat folded into the gather."""
    g = torch.Generator().manual_seed(9)
    a = quantize(torch.randn(2, 64, 10, 12, generator=g), dtype)
    bsk = quantize(torch.randn(2, 32, 20, 24, generator=g), dtype)
    wt = quantize(torch.randn(48, 96, 1, 1, generator=g) * 0.15, dtype)
    bias = torch.randn(48, generator=g) * 0.1
    pc = H.PackedConv(wt, bias, 1, 0, 1, True, dtype, device)
    y = H.conv2d(nhwc(a, dtype, device), pc, up2x=True, x2=nhwc(bsk, dtype, device))
    torch.cuda.synchronize()
    ref = F.silu(F.conv2d(torch.cat((F.interpolate(a, scale_factor=2.0, mode="nearest"), bsk), 1), wt, bias))
    check_close(back(y), ref, dtype, "dual-source up2x conv")
    # and a 3x3 through the upsample alone
    w3 = quantize(torch.randn(32, 64, 3, 3, generator=g) * 0.05, dtype)
    pc3 = H.PackedConv(w3, torch.zeros(32), 1, 1, 1, False, dtype, device, halo=False)
    y3 = H.conv2d(nhwc(a, dtype, device), pc3, up2x=True)
    torch.cuda.synchronize()
    check_close(back(y3), F.conv2d(F.interpolate(a, scale_factor=2.0, mode="nearest"), w3, None, 1, 1), dtype, "up2x 3x3 conv")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 36, 44), (3, 4, 4), (1, 132, 68), (2, 640, 640)],
                         ids=["64x96", "36x44 partial tiles", "4x4", "132x68", "640"])
def test_fused_stem2_matches_cpu(shape, dtype, device):
    """dy_stem2_fused: image -> SiLU(conv s2) -> round -> SiLU(conv s2), vs the two F.conv2d with the same rounding points;
    and bit-identical to the two-kernel path it replaces (same operands, same accumulation order per output)."""
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 3 + w)
    img = torch.rand(n, 3, h, w, generator=g)
    w0 = quantize(torch.randn(32, 3, 3, 3, generator=g) * 0.4, dtype)
    b0 = torch.randn(32, generator=g) * 0.2
    w1 = quantize(torch.randn(64, 32, 3, 3, generator=g) * (2.0 / 288) ** 0.5, dtype)
    b1 = torch.randn(64, generator=g) * 0.2
    assert H.stem2_fused_supported(3, 32, 64, h, w, dtype) and not H.stem2_fused_supported(3, 32, 64, h + 2, w, dtype)
    ps = H.PackedStem2(w0, b0, True, w1, b1, True, dtype, device)
    out = torch.full((n, h // 4, w // 4, 80), 3.0, dtype=dtype, device=device).permute(0, 3, 1, 2)  # a slice of a wider buffer
    y = H.stem2_fused(img.to(device), ps, out=out[:, 8:72])
    torch.cuda.synchronize()
    mid = quantize(F.silu(F.conv2d(quantize(img, dtype), w0, b0, 2, 1)), dtype)
    ref = F.silu(F.conv2d(mid, w1, b1, 2, 1))
    assert tuple(y.shape) == tuple(ref.shape)
    check_close(back(y), ref, dtype, f"stem2 {shape}", extra=1.5)
    assert float((out[:, :8].float() - 3.0).abs().max()) == 0.0 and float((out[:, 72:].float() - 3.0).abs().max()) == 0.0
    t = H.stem_conv(img.to(device), ps.stem)
    y2 = H.conv2d(t, H.PackedConv(w1, b1, 2, 1, 1, True, dtype, device))
    torch.cuda.synchronize()
    d = (back(y) - back(y2)).abs()
    assert float(d.max()) <= RTOL[dtype] * float(ref.abs().max()), "fused and layer-by-layer paths disagree beyond one rounding"
    assert float((d > 0).float().mean()) < 0.02, "fused path should reproduce the layer-by-layer result almost everywhere"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 3, 64, 96, 32), (1, 3, 50, 70, 16), (3, 3, 33, 31, 48), (1, 1, 40, 40, 80), (2, 3, 640, 640, 32)],
                         ids=["64x96->32", "odd 50x70->16", "odd 33x31->48", "cin1 ->80", "640 ->32"])
def test_fused_stem_matches_cpu(shape, dtype, device):
    """dy_stem_conv3x3s2_nchw: fp32 NCHW image -> SiLU(conv3x3 s2 + bias) NHWC, vs F.conv2d on the rounded inputs."""
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(h * 7 + cout)
    img = torch.rand(n, cin, h, w, generator=g)
    wt = quantize(torch.randn(cout, cin, 3, 3, generator=g) * 0.4, dtype)
    bias = torch.randn(cout, generator=g) * 0.2
    ps = H.PackedStem(wt, bias, True, dtype, device)
    y = H.stem_conv(img.to(device), ps)
    torch.cuda.synchronize()
    ref = F.silu(F.conv2d(quantize(img, dtype), wt, bias, 2, 1))
    assert tuple(y.shape) == tuple(ref.shape)
    check_close(back(y), ref, dtype, f"stem {shape}")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_grouped_conv(dtype, device):
    g = torch.Generator().manual_seed(11)
    x = quantize(torch.randn(2, 32, 12, 12, generator=g), dtype)
    wt = quantize(torch.randn(16, 2, 3, 3, generator=g) * 0.3, dtype)
    bias = torch.randn(16, generator=g) * 0.1
    pc = H.PackedConv(wt, bias, 2, 1, 16, True, dtype, device)
    y = H.conv2d(nhwc(x, dtype, device), pc)
    torch.cuda.synchronize()
    check_close(back(y), F.silu(F.conv2d(x, wt, bias, 2, 1, 1, 16)), dtype, "DWConv g=16")
    assert H.last_kernel_name() == "conv_smallgroup_kernel"  # r05: the -sf YAML's DWConv shape (2 input channels per output channel) has its own kernel


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", [(16, 8, 3, 2, 2, 40, 36), (64, 64, 3, 1, 1, 13, 17), (128, 32, 3, 2, 2, 24, 20), (24, 12, 3, 2, 1, 9, 11), (32, 16, 1, 1, 1, 8, 8)],
                         ids=["-sf layer 19 (n): 16->8 s2", "depthwise 64 s1", "cpg 4", "cout 12: generic kernel", "1x1 grouped"])
def test_small_group_conv_shapes(case, dtype, device):
    """DWConv(c1, c2, k, s) = Conv with g = gcd(c1, c2) (reference conv.py:102-107): one output channel per group with 1 / 2 / 4 input channels
    runs on conv_smallgroup_kernel where the layer is whole 16-byte chunks, anything else on the generic grouped kernel — all against F.conv2d."""
    import math

    c1, c2, k, s, b, h, w = case
    g = torch.Generator().manual_seed(c1 * 7 + c2)
    groups = math.gcd(c1, c2)
    x = quantize(torch.randn(b, c1, h, w, generator=g), dtype)
    wt = quantize(torch.randn(c2, c1 // groups, k, k, generator=g) * 0.3, dtype)
    bias = torch.randn(c2, generator=g) * 0.1
    pc = H.PackedConv(wt, bias, s, k // 2, groups, True, dtype, device)
    y = H.conv2d(nhwc(x, dtype, device), pc)
    torch.cuda.synchronize()
    check_close(back(y), F.silu(F.conv2d(x, wt, bias, s, k // 2, 1, groups)), dtype, f"grouped {case}")
    fast = c2 == groups and (c1 // groups) in (1, 2, 4) and c2 % H.elems_per_chunk(dtype) == 0
    assert (H.last_kernel_name() == "conv_smallgroup_kernel") == fast, H.last_kernel_name()


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_layout_upsample_copy_sppf(dtype, device):
    g = torch.Generator().manual_seed(13)
    img = torch.rand(3, 3, 32, 48, generator=g)
    x = H.to_nhwc(img.to(device), dtype)
    torch.cuda.synchronize()
    epc = H.elems_per_chunk(dtype)
    assert x.shape[1] == epc and torch.equal(back(x[:, :3]), quantize(img, dtype)) and float(back(x[:, 3:]).abs().max()) == 0
    assert torch.equal(H.to_nchw_f32(x[:, :3]).cpu(), quantize(img, dtype))
    t = quantize(torch.randn(2, 16, 6, 5, generator=g), dtype)
    up = H.upsample2x(nhwc(t, dtype, device, ld=32, c_off=16))
    torch.cuda.synchronize()
    assert torch.equal(back(up), F.interpolate(t, scale_factor=2.0, mode="nearest"))
    dst = torch.zeros((2, 6, 5, 48), dtype=dtype, device=device).permute(0, 3, 1, 2)
    H.copy_nhwc(nhwc(t, dtype, device), dst[:, 16:32])
    torch.cuda.synchronize()
    assert torch.equal(back(dst[:, 16:32]), t) and float(back(dst[:, :16]).abs().max()) == 0
    for (hh, ww) in ((20, 20), (7, 9), (40, 48)):
        s = quantize(torch.randn(2, 16, hh, ww, generator=g), dtype)
        buf = torch.zeros((2, hh, ww, 64), dtype=dtype, device=device).permute(0, 3, 1, 2)
        buf[:, :16] = s.to(dtype).to(device)
        H.sppf_maxpool3(buf[:, :16], buf[:, 16:32], buf[:, 32:48], buf[:, 48:64], 5)
        torch.cuda.synchronize()
        p = s
        for i in range(3):
            p = F.max_pool2d(p, 5, 1, 2)
            assert torch.equal(back(buf[:, 16 * (i + 1) : 16 * (i + 2)]), p), f"sppf pass {i} {hh}x{ww}"


def test_detect_decode_matches_oracle(device):
    g = torch.Generator().manual_seed(17)
    nc = 10
    shapes = [(16, 12), (8, 6), (4, 3), (2, 2)]
    strides = [4.0, 8.0, 16.0, 32.0]
    feats = [torch.randn(3, 64 + nc, h, w, generator=g) * 2.5 for h, w in shapes]
    ref = O.detect_decode(feats, strides, nc)
    dev = [nhwc(f, torch.float32, device, ld=76) for f in feats]
    y = H.detect_decode(dev, strides, nc, 16)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(ref.shape)
    assert torch.allclose(y.cpu(), ref, rtol=1e-5, atol=2e-4), float((y.cpu() - ref).abs().max())
    gp = golden("per_op.npz")
    raws = [torch.from_numpy(gp["det_raw0"]), torch.from_numpy(gp["det_raw1"])]
    yg = H.detect_decode([nhwc(r, torch.float32, device, ld=72) for r in raws], [8.0, 16.0], 5, 16)
    assert torch.allclose(yg.cpu(), torch.from_numpy(gp["det_y"]), rtol=1e-5, atol=2e-3)


def test_decode_fused_filter_equals_separate_filter(device):
    """dy_detect_decode with the fused candidate filter + dy_nms(prefiltered) == plain dy_nms on the same output."""
    g = torch.Generator().manual_seed(23)
    nc = 10
    shapes = [(40, 36), (20, 18), (10, 9), (5, 5)]  # 1440 + 360 + 90 + 25 anchors: partial 256-anchor tiles at every level
    strides = [4.0, 8.0, 16.0, 32.0]
    feats = [torch.randn(3, 64 + nc, h, w, generator=g) * 2.0 for h, w in shapes]
    dev = [nhwc(f, torch.float32, device, ld=76) for f in feats]
    A = sum(h * w for h, w in shapes)
    mask = torch.ones(nc, dtype=torch.uint8)
    mask[3] = 0
    mask = mask.to(device)
    for cm in (None, mask):
        bufs = H.NmsBuffers(3, A, 300, device)
        y = H.detect_decode(dev, strides, nc, 16, nms_bufs=bufs, conf_thres=0.3, classes_mask=cm)
        fused = H.nms(y, 0.3, 0.6, bufs=bufs, prefiltered=True, classes_mask=cm)
        torch.cuda.synchronize()
        f_out, f_cnt, f_idx = fused.out.clone(), fused.count.clone(), fused.index.clone()
        y2 = H.detect_decode(dev, strides, nc, 16)
        plain = H.nms(y2, 0.3, 0.6, classes_mask=cm)
        torch.cuda.synchronize()
        assert torch.equal(y, y2) and int(f_cnt.sum()) > 0
        assert torch.equal(f_cnt, plain.count) and torch.equal(f_out, plain.out) and torch.equal(f_idx, plain.index)
        assert torch.allclose(y.cpu(), O.detect_decode(feats, strides, nc), rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("nc,c_cls", [(10, 64), (80, 128), (17, 64)])
def test_fused_head_tail_decode(nc, c_cls, dtype, device):
    """dy_detect_head_decode (branch-tail 1x1 convs + decode + filter in one launch) against the oracle's
    conv2d -> Detect decode on the same (dtype-rounded) operands, and against the unfused device calls."""
    if dtype == torch.float32 and c_cls == 128:
        c_cls = 80  # the fp32 build holds at most 5 k-groups of 16 channels per branch in registers
    g = torch.Generator().manual_seed(31 + nc)
    shapes = [(40, 36), (20, 20), (10, 9), (5, 5), (1, 1)]  # 1440/400/90/25/1 anchors: several tiles, ragged tails, one anchor
    strides = [4.0, 8.0, 16.0, 32.0, 64.0]
    n = 3
    rnd = lambda t: t.to(dtype).float()
    xb = [rnd(torch.randn(n, 64, h, w, generator=g)) for h, w in shapes]
    xc = [rnd(torch.randn(n, c_cls, h, w, generator=g)) for h, w in shapes]
    wb = [rnd(torch.randn(64, 64, 1, 1, generator=g) * 0.25) for _ in shapes]
    wc = [rnd(torch.randn(nc, c_cls, 1, 1, generator=g) * 0.2) for _ in shapes]
    bb = [torch.randn(64, generator=g) for _ in shapes]
    bc = [torch.randn(nc, generator=g) - 1.0 for _ in shapes]
    feats = [torch.cat([F.conv2d(xb[i].double(), wb[i].double(), bb[i].double()), F.conv2d(xc[i].double(), wc[i].double(), bc[i].double())], 1).float()
             for i in range(len(shapes))]
    ref = O.detect_decode(feats, strides, nc)
    assert H.head_decode_supported(64, c_cls, nc, 16, dtype)
    dxb = [nhwc(t, dtype, device) for t in xb]
    dxc = [nhwc(t, dtype, device) for t in xc]
    pb = [H.pack_frag1x1(wb[i], bb[i], dtype, device) for i in range(len(shapes))]
    pc = [H.pack_frag1x1(wc[i], bc[i], dtype, device) for i in range(len(shapes))]
    A = sum(h * w for h, w in shapes)
    bufs = H.NmsBuffers(n, A, 300, device)
    y = H.detect_head_decode(dxb, dxc, pb, pc, strides, nc, 16, nms_bufs=bufs, conf_thres=0.3)
    kept = H.nms(y, 0.3, 0.6, bufs=bufs, prefiltered=True)
    torch.cuda.synchronize()
    f_out, f_cnt, f_idx = kept.out.clone(), kept.count.clone(), kept.index.clone()
    # operands are exactly representable, products accumulate in fp32 on the device: only summation order differs
    assert torch.allclose(y.cpu(), ref, rtol=1e-4, atol=2e-3), float((y.cpu() - ref).abs().max())
    plain = H.nms(y, 0.3, 0.6)
    torch.cuda.synchronize()
    assert int(f_cnt.sum()) > 0 and torch.equal(f_cnt, plain.count) and torch.equal(f_out, plain.out) and torch.equal(f_idx, plain.index)


def test_fused_head_unsupported_shapes_are_refused(device):
    assert not H.head_decode_supported(64, 80, 80, 16, torch.bfloat16)  # 80 channels: not whole 32-channel k-groups
    assert not H.head_decode_supported(32, 64, 10, 16, torch.bfloat16)
    assert not H.head_decode_supported(64, 64, 10, 8, torch.bfloat16)
    assert H.head_decode_supported(64, 80, 80, 16, torch.float32)


def _run_nms(pred, device, **kw):
    kw = dict(kw)
    classes = kw.pop("classes", None)
    mask = None
    nc = pred.shape[1] - 4
    if classes is not None:
        mask = torch.zeros(nc, dtype=torch.uint8)
        mask[classes] = 1
        mask = mask.to(device)
    b = H.nms(pred.to(device).contiguous(), kw.get("conf_thres", 0.25), kw.get("iou_thres", 0.45), max_det=kw.get("max_det", 300),
              max_nms=kw.get("max_nms", 30000), agnostic=kw.get("agnostic", False), classes_mask=mask, multi_label=kw.get("multi_label", False))
    torch.cuda.synchronize()
    counts = b.count.cpu().tolist()
    return [b.out[i, :c].cpu() for i, c in enumerate(counts)], [b.index[i, :c].cpu() for i, c in enumerate(counts)], b


def test_nms_golden_cases_bit_exact(device):
    g = golden("nms.npz")
    for name in sorted({k.split("__")[0] for k in g.files}):
        pred = torch.from_numpy(g[f"{name}__pred"])
        kw = ast.literal_eval(str(g[f"{name}__kw"]))
        rows, _, bufs = _run_nms(pred, device, **kw)
        exp = split_rows(g[f"{name}__out"], g[f"{name}__n"])
        assert [len(r) for r in rows] == [len(e) for e in exp], f"{name}: counts {[len(r) for r in rows]} vs {[len(e) for e in exp]}"
        for r, e in zip(rows, exp):
            assert np.array_equal(r.numpy(), e), f"{name}: kept rows differ"
        md = bufs.max_det
        for i, c in enumerate(bufs.count.cpu().tolist()):
            assert float(bufs.out[i, c:].abs().sum()) == 0 and (c == md or int(bufs.index[i, c]) == -1)


def test_nms_multilabel_golden_cases_bit_exact(device):
    """The validator's NMS (``multi_label=True``: one candidate per (anchor, class) pair above conf, utils/ops.py:286-288; call site
    models/yolo/detect/val.py:93-106) against the rows the REAL reference's non_max_suppression returned (tests/golden/nms_ml.npz):
    rows bit-exact, kept ANCHOR indices identical — conf 0.001 crowds, agnostic, class filter, max_det, more candidates than max_nms,
    several labels on one anchor, and nc = 1 (where the reference switches multi_label off)."""
    g = golden("nms_ml.npz")
    names = sorted({k.split("__")[0] for k in g.files})
    assert len(names) >= 8
    for name in names:
        pred = torch.from_numpy(g[f"{name}__pred"])
        kw = ast.literal_eval(str(g[f"{name}__kw"]))
        rows, idx, bufs = _run_nms(pred, device, **kw)
        exp = split_rows(g[f"{name}__out"], g[f"{name}__n"])
        assert [len(r) for r in rows] == [len(e) for e in exp], f"{name}: counts {[len(r) for r in rows]} vs {[len(e) for e in exp]}"
        for r, e in zip(rows, exp):
            assert np.array_equal(r.numpy(), e), f"{name}: kept rows differ"
        assert np.array_equal(torch.cat(idx).numpy().astype(np.int64), g[f"{name}__idx"]), f"{name}: kept anchor indices differ"


def test_nms_multilabel_random_matches_oracle(device):
    """34,000 anchors x 10 classes at conf 0.001 (the validator's setting): ~300k (anchor, class) candidates per image through the
    global-memory sort and the max_nms cut; rows and anchor indices bit-exact vs the oracle (continuous scores: no ties at the cut)."""
    g = torch.Generator().manual_seed(99)
    batch, n_anchors, nc = 2, 34000, 10
    xy = torch.rand(batch, 2, n_anchors, generator=g) * 600 + 20
    wh = torch.rand(batch, 2, n_anchors, generator=g) * 80 + 2
    # DISTINCT scores (a permutation of k / N): among 340,000 candidates per image random floats tie, and the reference cuts to max_nms
    # with an unstable argsort (ops.py:302), which leaves the order of ties — and with it the kept set — undefined
    N = nc * n_anchors
    sc = torch.stack([(torch.randperm(N, generator=g).float() + 0.5) / N for _ in range(batch)]).view(batch, nc, n_anchors)
    pred = torch.cat((xy, wh, sc), 1).contiguous()
    exp, exp_idx = O.non_max_suppression(pred, 0.001, 0.7, max_det=300, nc=nc, max_nms=30000, return_index=True, multi_label=True)
    rows, idx, _ = _run_nms(pred, device, conf_thres=0.001, iou_thres=0.7, max_det=300, max_nms=30000, multi_label=True)
    for i in range(batch):
        assert torch.equal(idx[i].long(), exp_idx[i]), f"image {i}: kept anchor indices differ"
        assert torch.equal(rows[i], exp[i]), f"image {i}: rows differ"


@pytest.mark.parametrize("n_anchors,batch,conf", [(34000, 3, 0.25), (2100, 5, 0.05), (40, 7, 0.2), (34000, 2, 0.001)])
def test_nms_random_matches_oracle_indices(n_anchors, batch, conf, device):
    """Indices/classes bit-exact and rows identical vs the oracle; conf=0.001 drives ~30k candidates
    per image through the global-memory sort path and the max_nms truncation."""
    g = torch.Generator().manual_seed(n_anchors + batch)
    nc = 10
    xy = torch.rand(batch, 2, n_anchors, generator=g) * 600 + 20
    wh = torch.rand(batch, 2, n_anchors, generator=g) * 80 + 2
    sc = torch.rand(batch, nc, n_anchors, generator=g) ** (8 if conf > 0.01 else 1)
    if conf > 0.01:
        sc = torch.round(sc * 512) / 512  # coarse scores: plenty of exact ties for the stable-order rule
    pred = torch.cat((xy, wh, sc), 1).contiguous()
    max_nms = 30000 if conf > 0.01 else 20000
    exp, exp_idx = O.non_max_suppression(pred, conf, 0.7, max_det=300, nc=nc, max_nms=max_nms, return_index=True)
    rows, idx, _ = _run_nms(pred, device, conf_thres=conf, iou_thres=0.7, max_det=300, max_nms=max_nms)
    for i in range(batch):
        # (conf=0.001 keeps continuous scores: the reference truncates to max_nms with an UNSTABLE argsort,
        #  ops.py:302, so tie order is undefined there; without ties the result is unique)
        assert torch.equal(idx[i].long(), exp_idx[i]), f"image {i}: kept anchor indices differ"
        assert torch.equal(rows[i], exp[i]), f"image {i}: rows differ"


def test_nms_public_api_list_and_errors(device):
    from drone_yolo_amd.utils import ops

    g = golden("nms.npz")
    pred = torch.from_numpy(g["crowd__pred"]).to(device)
    out = ops.non_max_suppression(pred, 0.25, 0.7)
    exp = split_rows(g["crowd__out"], g["crowd__n"])
    assert all(np.array_equal(o.cpu().numpy(), e) for o, e in zip(out, exp))
    with pytest.raises(AssertionError):
        ops.non_max_suppression(pred, 1.5, 0.7)
    ml = ops.non_max_suppression(pred, 0.25, 0.7, multi_label=True)  # the validator's form: at least the single-label detections' count
    assert len(ml) == len(out) and all(len(a) >= len(b) or len(a) == 300 for a, b in zip(ml, out))
    with pytest.raises(NotImplementedError):
        ops.non_max_suppression(pred, 0.25, 0.7, rotated=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.non_max_suppression(pred.cpu(), 0.25, 0.7)


def test_scale_boxes_kernel(device):
    b = H.NmsBuffers(2, 64, 5, device)
    rows = torch.tensor([[[-5.0, 10, 700, 500, 0.9, 1], [30, 40, 50, 60, 0.8, 2], [600, 300, 650, 490, 0.7, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]]] * 2)
    b.out.copy_(rows)
    b.count.copy_(torch.tensor([3, 1], dtype=torch.int32))
    gain, px, py = 0.8, 0.0, 0.0
    params = torch.tensor([[gain, px, py, 800.0, 480.0]] * 2, device=device)
    H.scale_boxes_(b, params)
    torch.cuda.synchronize()
    exp = O.scale_boxes((384, 640), rows[0, :3, :4].clone(), (480, 800))
    assert torch.allclose(b.out[0, :3, :4].cpu(), exp, atol=1e-4)
    assert torch.equal(b.out[1, 1:].cpu(), rows[1, 1:]) and torch.equal(b.out[0, :, 4:].cpu(), rows[0, :, 4:])


# ---- image sources: LetterBox + BGR->RGB + CHW + /255 ------------------------------------------------------------------


@pytest.mark.parametrize("shape,imgsz,auto", [((1080, 1920), 640, True), ((480, 640), 640, True), ((640, 640), 640, False), ((333, 517), 640, False),
                                               ((97, 1201), (320, 640), True), ((720, 1280), 1280, True)])
def test_letterbox_kernel_bit_exact(shape, imgsz, auto, device):
    """dy_letterbox_u8_to_nchw_f32 against oracle/letterbox_oracle.py (LetterBox + 8-bit bilinear + preprocess): identical."""
    from drone_yolo_amd.data.augment import LetterBox
    from oracle import letterbox_oracle as LB

    rng = np.random.default_rng(shape[0])
    frames = [rng.integers(0, 256, (*shape, 3), dtype=np.uint8) for _ in range(2)]
    new_shape = (imgsz, imgsz) if isinstance(imgsz, int) else imgsz
    ref = LB.preprocess(frames, new_shape, auto=auto, stride=32)
    lb = LetterBox(new_shape, auto=auto, stride=32)
    assert lb.geometry(shape) == LB.letterbox_geometry(shape, new_shape, auto=auto, stride=32)
    out = lb(torch.from_numpy(np.stack(frames)).to(device))
    torch.cuda.synchronize()
    assert tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy(), ref), float(np.abs(out.cpu().numpy() - ref).max()) * 255


# ---- fused C2f block -------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape,shortcut", [((2, 64, 40, 36), True), ((3, 64, 16, 16), False), ((1, 64, 33, 50), True), ((2, 64, 160, 160), True)])
def test_c2f_fused_block_matches_cpu_chain_and_layerwise(shape, shortcut, dtype, device):
    """dy_c2f_fused against the CPU chain conv -> round -> conv ... with every intermediate rounded to the storage dtype
    (what both the layer-by-layer device path and the fused kernel do), and against the layer-by-layer device path."""
    from drone_yolo_amd.nn.modules import C2f

    g = torch.Generator().manual_seed(shape[2])
    blk = C2f(64, 64, n=1, shortcut=shortcut).eval()
    for prm in blk.parameters():
        prm.data = torch.randn(prm.shape, generator=g) * (0.12 if prm.dim() > 1 else 0.3) + (1.0 if prm.dim() == 1 else 0.0)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.eps = 1e-3
    x = quantize(torch.randn(shape, generator=g), dtype)
    q = lambda t: quantize(t, dtype)  # noqa: E731
    from drone_yolo_amd.nn.modules.conv import fold_conv_bn

    def cba(conv, t, k):
        w, b = fold_conv_bn(conv.conv.weight, None, conv.bn)
        return q(F.silu(F.conv2d(t, q(w), b, 1, k // 2)))

    y = cba(blk.cv1, x, 1)
    y0, y1 = y[:, :32], y[:, 32:]
    t = cba(blk.m[0].cv1, y1, 3)
    w, b = fold_conv_bn(blk.m[0].cv2.conv.weight, None, blk.m[0].cv2.bn)
    y2 = F.silu(F.conv2d(t, q(w), b, 1, 1))
    y2 = q(y2 + y1) if shortcut else q(y2)
    ref = cba(blk.cv2, torch.cat((y0, y1, y2), 1), 1)
    blk = blk.to(device)
    xd = nhwc(x, dtype, device, ld=64 + 16, c_off=8)
    blk.fuse_block = True
    got = blk(xd)
    blk.fuse_block = False
    layerwise = blk(xd)
    torch.cuda.synchronize()
    check_close(back(got), ref, dtype, "fused C2f vs CPU chain", extra=3.0)
    check_close(back(got), back(layerwise), dtype, "fused C2f vs layer-by-layer device path", extra=3.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape,shortcut", [((2, 40, 36), False), ((1, 34, 50), True), ((3, 16, 16), False), ((2, 160, 160), False)])
def test_c2f_fused_block_with_upsample_concat_matches_cpu_chain(shape, shortcut, dtype, device):
    """The neck's stride-4 block: nn.Upsample(2, 'nearest') + Concat + C2f(192, 64) in one dy_c2f_fused launch against the CPU
    chain on cat(upsample(x_lo), x) with every intermediate rounded to the storage dtype, and against the layer-by-layer device
    path (cv1 gathering both sources, two 3x3 launches, cv2)."""
    from drone_yolo_amd.nn.modules import C2f
    from drone_yolo_amd.nn.modules.conv import fold_conv_bn

    n, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    blk = C2f(192, 64, n=1, shortcut=shortcut).eval()
    for prm in blk.parameters():
        prm.data = torch.randn(prm.shape, generator=g) * (0.08 if prm.dim() > 1 else 0.3) + (1.0 if prm.dim() == 1 else 0.0)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.eps = 1e-3
    q = lambda t: quantize(t, dtype)  # noqa: E731
    x_lo, x_hi = q(torch.randn((n, 128, h // 2, w // 2), generator=g)), q(torch.randn((n, 64, h, w), generator=g))
    x = torch.cat((F.interpolate(x_lo, scale_factor=2, mode="nearest"), x_hi), 1)

    def cba(conv, t, k):
        wt, b = fold_conv_bn(conv.conv.weight, None, conv.bn)
        return q(F.silu(F.conv2d(t, q(wt), b, 1, k // 2)))

    y = cba(blk.cv1, x, 1)
    y0, y1 = y[:, :32], y[:, 32:]
    t = cba(blk.m[0].cv1, y1, 3)
    wt, b = fold_conv_bn(blk.m[0].cv2.conv.weight, None, blk.m[0].cv2.bn)
    y2 = F.silu(F.conv2d(t, q(wt), b, 1, 1))
    y2 = q(y2 + y1) if shortcut else q(y2)
    ref = cba(blk.cv2, torch.cat((y0, y1, y2), 1), 1)
    blk = blk.to(device)
    lo_d, hi_d = nhwc(x_lo, dtype, device, ld=128 + 8, c_off=8), nhwc(x_hi, dtype, device, ld=64 + 16, c_off=8)
    assert H.c2f_fused_supported(192, 32, 64, 1, dtype, cin_lo=128)
    blk.fuse_block = True
    got = blk(lo_d, x2=hi_d, up2x=True)
    blk.fuse_block = False
    layerwise = blk(lo_d, x2=hi_d, up2x=True)
    torch.cuda.synchronize()
    check_close(back(got), ref, dtype, "fused upsample + concat + C2f vs CPU chain", extra=3.0)
    check_close(back(got), back(layerwise), dtype, "fused upsample + concat + C2f vs layer-by-layer device path", extra=3.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_detect_stacked_first_convs_match_separate(dtype, device):
    """Detect._trunks: cv2[i][0] / cv3[i][0] stacked into one conv on the deep levels == the two separate branches."""
    from drone_yolo_amd.nn.modules import Detect

    g = torch.Generator().manual_seed(5)
    class LegacyDetect(Detect):  # the v8 head of the Drone-YOLO YAMLs: two 3x3 convs per branch (nn/tasks.py sets legacy)
        legacy = True

    det = LegacyDetect(nc=10, ch=(64, 128, 256)).eval()
    for prm in det.parameters():
        prm.data = torch.randn(prm.shape, generator=g) * (0.05 if prm.dim() > 1 else 0.2) + (1.0 if prm.dim() == 1 else 0.0)
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    det = det.to(device)
    xs = [nhwc(quantize(torch.randn(2, c, s, s, generator=g), dtype), dtype, device) for c, s in ((64, 40), (128, 20), (256, 10))]
    assert det._packed_first(0, dtype, device) is not None and det._packed_first(1, dtype, device) is not None  # from 64 input channels up
    for i, x in enumerate(xs):
        det.fuse_first = True
        tb, tc = det._trunks(i, x)
        det.fuse_first = False
        sb, sc = det._trunks(i, x)
        torch.cuda.synchronize()
        check_close(back(tb), back(sb), dtype, f"stacked box trunk level {i}", extra=3.0)
        check_close(back(tc), back(sc), dtype, f"stacked class trunk level {i}", extra=3.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_detect_branch_fused_matches_tail_path(dtype, device):
    """dy_detect_branch_fused (second 3x3 conv + 1x1 + decode share + candidate filter of one branch in one kernel) against the
    layer-by-layer trunks + fused tail (dy_detect_head_decode) on the same 16-bit inputs: same operands and rounding points
    (the 3x3 output is rounded to the storage type in both), different summation orders -> decoded boxes / scores agree to a few
    roundings of the trunk activations, and the candidate lists hold the same anchors except where a score sits at conf.
    Ragged maps (23x37: tiles overhang) and a class mask are part of the case."""
    from drone_yolo_amd.nn.modules import Detect

    g = torch.Generator().manual_seed(11)

    class LegacyDetect(Detect):
        legacy = True

    det = LegacyDetect(nc=10, ch=(64, 128)).eval()
    for prm in det.parameters():
        prm.data = torch.randn(prm.shape, generator=g) * (0.05 if prm.dim() > 1 else 0.2) + (1.0 if prm.dim() == 1 else 0.0)
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    det = det.to(device)
    det.stride = torch.tensor([4.0, 8.0])
    xs = [nhwc(quantize(torch.randn(3, c, h, w, generator=g), dtype), dtype, device) for c, h, w in ((64, 23, 37), (128, 12, 19))]
    A = 23 * 37 + 12 * 19
    mask = torch.ones(10, dtype=torch.uint8)
    mask[3] = 0
    outs = []
    for fuse_branch in (True, False):
        det.fuse_branch, det.fuse_tail = fuse_branch, True
        holder = {}

        def make_bufs(nb, anchors):
            holder["b"] = H.NmsBuffers(nb, anchors, 300, device)
            return holder["b"]

        det.fused_nms = (make_bufs, 0.5, mask.to(device))
        assert det._branches_fusable(dtype) == fuse_branch
        y, _ = det(xs)
        bufs = H.nms(y, 0.5, 0.7, max_det=300, nc=10, classes_mask=mask.to(device), bufs=holder["b"], prefiltered=True)
        torch.cuda.synchronize()
        outs.append((y.cpu(), bufs.count.cpu().clone(), bufs.index.cpu().clone(), bufs.out.cpu().clone()))
    det.fused_nms = None
    (y1, c1, i1, o1), (y0, c0, i0, o0) = outs
    assert tuple(y1.shape) == (3, 14, A) and bool(torch.isfinite(y1).all())
    tol = 0.06 if dtype == torch.bfloat16 else 0.01
    assert float((y1[:, :4] - y0[:, :4]).abs().max()) <= tol * float(y0[:, :4].abs().max()), float((y1[:, :4] - y0[:, :4]).abs().max())
    assert float((y1[:, 4:] - y0[:, 4:]).abs().max()) <= tol
    for b in range(3):  # kept detections: same anchors except score-at-threshold / near-tie cases
        s1, s0 = set(i1[b, : int(c1[b])].tolist()), set(i0[b, : int(c0[b])].tolist())
        assert len(s1 ^ s0) <= max(2, int(0.05 * len(s0))), (b, len(s1), len(s0), len(s1 ^ s0))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shapes", [((64, 23, 37), (128, 12, 19)), ((64, 8, 16), (64, 5, 3)), ((64, 40, 40), (256, 9, 11))], ids=["ragged", "tiny", "square+deep"])
def test_detect_branch_fused_matches_cpu_chain(shapes, dtype, device):
    """Localising test: dy_detect_branch_fused — second 3x3 conv + SiLU, 1x1 to 4*reg_max box bins / nc class
    logits, DFL softmax-expectation + dist2bbox / sigmoid — against an fp32 CPU chain on the SAME quantised operands:
    F.conv2d -> SiLU -> round to the storage type (the trunk activations' rounding points, head.py:64-72) -> 1x1 in fp32 ->
    the oracle's Detect decode (head.py:100-131, tal.py:333-363).  No other HIP path is involved in the expectation, so a
    defect points at this kernel (or at the trunk's first conv, which the CPU chain restates too)."""
    from drone_yolo_amd.nn.modules import Detect
    from drone_yolo_amd.nn.modules.conv import fold_conv_bn

    g = torch.Generator().manual_seed(17 + shapes[0][1])

    class LegacyDetect(Detect):
        legacy = True

    nc = 10
    det = LegacyDetect(nc=nc, ch=tuple(s[0] for s in shapes)).eval()
    for prm in det.parameters():
        prm.data = torch.randn(prm.shape, generator=g) * (0.05 if prm.dim() > 1 else 0.2) + (1.0 if prm.dim() == 1 else 0.0)
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    strides = [4.0, 8.0]
    q = lambda t: quantize(t, dtype)  # noqa: E731
    xs_cpu = [q(torch.randn(3, c, h, w, generator=g)) for c, h, w in shapes]

    def cba(conv, t):
        wt, b = fold_conv_bn(conv.conv.weight.detach(), None, conv.bn)
        return q(F.silu(F.conv2d(t, q(wt), b, 1, 1)))

    feats = []
    with torch.no_grad():
        for i, x in enumerate(xs_cpu):
            tb = cba(det.cv2[i][1], cba(det.cv2[i][0], x))
            tc = cba(det.cv3[i][1], cba(det.cv3[i][0], x))
            lb = F.conv2d(tb.double(), q(det.cv2[i][2].weight.detach()).double(), det.cv2[i][2].bias.detach().double()).float()
            lc = F.conv2d(tc.double(), q(det.cv3[i][2].weight.detach()).double(), det.cv3[i][2].bias.detach().double()).float()
            feats.append(torch.cat((lb, lc), 1))
    ref = O.detect_decode(feats, strides, nc)
    det = det.to(device)
    det.stride = torch.tensor(strides)
    xs = [nhwc(t, dtype, device) for t in xs_cpu]
    det.fuse_branch, det.fuse_tail = True, True
    holder = {}

    def make_bufs(nb, anchors):
        holder["b"] = H.NmsBuffers(nb, anchors, 300, device)
        return holder["b"]

    det.fused_nms = (make_bufs, 0.5, None)
    assert det._branches_fusable(dtype)
    y, _ = det(xs)
    torch.cuda.synchronize()
    det.fused_nms = None
    y = y.cpu()
    assert tuple(y.shape) == tuple(ref.shape) and bool(torch.isfinite(y).all())
    # same operands, same rounding points; only fp32 summation order (and rare 1-ulp flips of a trunk activation) differ
    box_tol = (0.03 if dtype == torch.bfloat16 else 0.004) * float(ref[:, :4].abs().max())
    cls_tol = 0.03 if dtype == torch.bfloat16 else 0.004
    assert float((y[:, :4] - ref[:, :4]).abs().max()) <= box_tol, (float((y[:, :4] - ref[:, :4]).abs().max()), box_tol)
    assert float((y[:, 4:] - ref[:, 4:]).abs().max()) <= cls_tol, float((y[:, 4:] - ref[:, 4:]).abs().max())
    # the candidate filter inside the class branch: every anchor whose reference score clears conf by a margin is listed
    bufs = H.nms(y.to(device), 0.5, 0.7, max_det=300, nc=nc, bufs=holder["b"], prefiltered=True)
    plain = H.nms(y.to(device), 0.5, 0.7, max_det=300, nc=nc)
    torch.cuda.synchronize()
    assert torch.equal(bufs.count, plain.count) and torch.equal(bufs.index, plain.index)


@pytest.mark.parametrize("case", [(64, 64, 3, 1, 2, 24, 20, True), (160, 80, 1, 1, 2, 17, 19, True), (80, 160, 3, 2, 2, 24, 28, True), (640, 320, 1, 1, 1, 12, 12, False),
                                  (96, 48, 3, 1, 3, 10, 10, True), (160, 160, 3, 1, 3, 33, 29, True), (320, 320, 3, 1, 1, 16, 16, True), (400, 160, 1, 1, 2, 24, 20, True)],
                         ids=["3x3 64->64", "1x1 160->80", "3x3 s2 80->160", "1x1 640->320 no act", "3x3 96->48", "3x3 160->160", "3x3 320->320", "1x1 400->160"])
def test_fp8_conv_matches_dequantised_reference(case, device):
    """DY_FP8 (BASELINE config 5): e4m3fn activations and per-output-channel-scaled e4m3fn weights on the fp8 MFMA, fp32 accumulate.
    The reference is the SAME arithmetic on the CPU: dequantise the fp8 input and the fp8 weights exactly, convolve in fp32.
    (a) fp32 output (out_f32, the Detect logits' form): equal to accumulation order, 1e-4 of the output scale; (b) fp8 output:
    the reference rounded to e4m3 at the same activation scale, equal up to one quantum where the fp32 sums straddle a rounding
    boundary (<= 2 % of the elements); (c) a residual (Bottleneck shortcut) is added in real units before the rounding."""
    import torch.nn.functional as F

    cin, cout, k, s_, b, h, w, act = case
    FP8 = H.FP8
    g = torch.Generator().manual_seed(cin * 7 + cout)
    act_scale = 0.05
    H.set_fp8_act_scale(act_scale)
    try:
        xq = (torch.randn(b, cin, h, w, generator=g) * 1.5 / act_scale).to(FP8)
        wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.2
        pc = H.PackedConv(wt, bias, s_, k // 2, 1, act, FP8, device)
        ws = (wt.reshape(cout, -1).abs().amax(1) / 448.0).clamp_min(1e-12)
        wq = (wt / ws.view(-1, 1, 1, 1)).to(FP8).float() * ws.view(-1, 1, 1, 1)  # what the device multiplies with, exactly
        z = F.conv2d(xq.float() * act_scale, wq, bias, s_, k // 2)
        ref = F.silu(z) if act else z
        xd = xq.permute(0, 2, 3, 1).contiguous().to(device).permute(0, 3, 1, 2)
        y32 = H.conv2d(xd, pc, out_f32=True) if not act else None
        yq = H.conv2d(xd, pc)
        rq = (torch.randn(ref.shape, generator=g) / act_scale).to(FP8)
        yr = H.conv2d(xd, pc, residual=rq.permute(0, 2, 3, 1).contiguous().to(device).permute(0, 3, 1, 2))
        torch.cuda.synchronize()
        scale = float(ref.abs().max())
        if y32 is not None:
            assert float((y32.cpu() - ref).abs().max()) <= 1e-4 * scale
        for got, want in ((yq, ref), (yr, ref + rq.float() * act_scale)):
            want_q = (want / act_scale).clamp(-448, 448).to(FP8).float()
            diff = (got.cpu().float() - want_q).abs()
            step = want_q.abs().clamp_min(2.0 ** -6) * 0.126  # one e4m3 quantum is <= 1/8 of the value (2^-9 below 2^-6)
            assert bool((diff <= step).all()), float((diff / step).max())
            assert float((diff > 0).float().mean()) <= 0.02
        # r04: fp8 convolutions run on the flat-K kernel's block-scaled MFMA (v_mfma_f32_16x16x128_f8f6f4), and the output type may differ
        # from the input's: fp8 -> float16 (a 16-bit Detect tail behind an fp8 trunk) is the fp32 result rounded once to float16
        y16 = H.conv2d(xd, pc, out_dtype=torch.float16)
        torch.cuda.synchronize()
        assert H.last_kernel_name().startswith("conv_gemm_fk_kernel"), H.last_kernel_name()
        assert y16.dtype == torch.float16 and float((y16.cpu().float() - ref).abs().max()) <= 1.2e-3 * scale
        # float16 -> fp8 (the hand-over into the trunk): 16-bit operands, the output quantised at the activation scale
        x16 = quantize(xq.float() * act_scale, torch.float16)
        w16 = quantize(wt, torch.float16)
        pc16 = H.PackedConv(w16, bias, s_, k // 2, 1, act, torch.float16, device, halo=False)
        y8 = H.conv2d(x16.permute(0, 2, 3, 1).contiguous().to(device, torch.float16).permute(0, 3, 1, 2), pc16, out_dtype=FP8)
        torch.cuda.synchronize()
        z16 = F.conv2d(x16, w16, bias, s_, k // 2)
        want_q = ((F.silu(z16) if act else z16) / act_scale).clamp(-448, 448).to(FP8).float()
        diff = (y8.cpu().float() - want_q).abs()
        assert y8.dtype == FP8 and bool((diff <= want_q.abs().clamp_min(2.0 ** -6) * 0.126).all()) and float((diff > 0).float().mean()) <= 0.02
    finally:
        H.set_fp8_act_scale(1.0)


def test_fp8_quantize_and_pool(device):
    """dy_quantize_fp8_nhwc == torch's e4m3fn cast of x / scale (round to nearest even, saturating); SPPF's three max pools are
    exact on fp8 (max commutes with the monotonic quantisation)."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 9, 11, generator=g) * 30
    x[0, 0, 0, 0], x[0, 1, 0, 0] = 1e4, -1e4  # saturate
    H.set_fp8_act_scale(0.25)
    try:
        x16 = x.half().permute(0, 2, 3, 1).contiguous().to(device).permute(0, 3, 1, 2)
        q = H.quantize_fp8(x16)
        torch.cuda.synchronize()
        want = (x.half().float() / 0.25).clamp(-448, 448).to(H.FP8)
        assert torch.equal(q.cpu().float(), want.float())
        y1, y2, y3 = (H.alloc_nhwc(2, 32, 9, 11, H.FP8, device) for _ in range(3))
        H.sppf_maxpool3(q, y1, y2, y3, 5)
        torch.cuda.synchronize()
        m1 = F.max_pool2d(want.float(), 5, 1, 2)
        m2 = F.max_pool2d(m1, 5, 1, 2)
        m3 = F.max_pool2d(m2, 5, 1, 2)
        assert torch.equal(y1.cpu().float(), m1) and torch.equal(y2.cpu().float(), m2) and torch.equal(y3.cpu().float(), m3)
    finally:
        H.set_fp8_act_scale(1.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
def test_device_weight_packing_equals_host_packing(dtype, device):
    """dy_pack_conv_weights (one launch: layout + zero padding + cast, and transpose + flip for the input-gradient convolution)
    writes exactly the bytes the host-side pad / permute / flip / cast chain of PackedConv produces, for every weight layout
    (ROWS incl. padded rows / columns, HALO3X3, FRAG1X1), for a strided (permuted) source and for a zero-padded input channel."""
    g = torch.Generator().manual_seed(2)
    cases = [(64, 64, 3, 1, None), (48, 40, 3, 1, None), (128, 256, 3, 2, None), (96, 192, 1, 1, None), (20, 100, 1, 1, None), (256, 128, 3, 1, None), (32, 3, 3, 2, 8),
             (16, 16, 3, 1, None)]
    for cout, cin, k, s_, cin_pad in cases:
        w = torch.randn(cout, cin, k, k, generator=g)
        b = torch.randn(cout, generator=g)
        host = H.PackedConv(w, b, s_, k // 2, 1, True, dtype, device, cin_pad=cin_pad)
        dev = H.PackedConv(w.to(device), b.to(device), s_, k // 2, 1, True, dtype, device, cin_pad=cin_pad)
        torch.cuda.synchronize()
        assert (host.layout, host.k_pad, host.cout_pad, host.cin, host.cout) == (dev.layout, dev.k_pad, dev.cout_pad, dev.cin, dev.cout), (cout, cin, k)
        assert host.w.numel() == dev.w.numel() and torch.equal(host.w.cpu().reshape(-1).view(torch.uint8), dev.w.cpu().reshape(-1).view(torch.uint8)), (cout, cin, k, host.layout)
        assert torch.equal(host.b.cpu(), dev.b.cpu())
        if cin_pad is None and cin >= 16:
            # a permuted (OHWI-stored) source and the transposed / flipped pack of the input-gradient convolution
            w_ohwi = w.permute(0, 2, 3, 1).contiguous().to(device).permute(0, 3, 1, 2)
            dev2 = H.PackedConv(w_ohwi, b.to(device), s_, k // 2, 1, True, dtype, device)
            hd = H.pack_dgrad(w, s_, dtype, device)
            dd = H.pack_dgrad(w_ohwi, s_, dtype, device)
            torch.cuda.synchronize()
            assert torch.equal(dev2.w.cpu().reshape(-1).view(torch.uint8), host.w.cpu().reshape(-1).view(torch.uint8))
            assert hd.layout == dd.layout and torch.equal(hd.w.cpu().reshape(-1).view(torch.uint8), dd.w.cpu().reshape(-1).view(torch.uint8)), (cout, cin, k, hd.layout)


@pytest.mark.parametrize("shape", [(2, 3, 64, 64, 96, 96), (1, 3, 64, 96, 32, 64), (2, 3, 50, 70, 64, 96), (1, 3, 640, 640, 928, 928)], ids=["up 1.5x", "down 0.5x", "odd aspect", "full size up"])
def test_multi_scale_resize_matches_torch_interpolate(shape, device):
    """dy_resize_bilinear_u8_nchw_f32 = the ``multi_scale`` branch of preprocess_batch (models/yolo/detect/train.py:60-73):
    nn.functional.interpolate(img.float() / 255, size, mode="bilinear", align_corners=False) — the reference's own op on the CPU — to fp32 round-off."""
    n, c, h, w, ho, wo = shape
    img = torch.randint(0, 256, (n, c, h, w), generator=torch.Generator().manual_seed(h + wo), dtype=torch.uint8)
    ref = F.interpolate(img.float() / 255, size=(ho, wo), mode="bilinear", align_corners=False)
    got = H.resize_bilinear_u8(img.to(device), (ho, wo))
    torch.cuda.synchronize()
    assert tuple(got.shape) == tuple(ref.shape) and float((got.cpu() - ref).abs().max()) <= 2e-6
