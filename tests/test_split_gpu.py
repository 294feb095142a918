"""GPU parity of the split-float16 storage type (DY_F16X2, include/dyolo.h): x ~= hi + lo * 2^-11, three 16-bit MFMAs per product.

The type exists to meet the north-star bar (class / index exact, IoU >= 0.999 against the fp32 CPU path) at 16-bit MFMA speed, so it is
held to fp32's own tolerances: kernels against a float64 CPU convolution to a few fp32 ulps of the output scale, whole models to the
SAME assertions the fp32 device path meets (kept sets, classes and order identical to the reference's rows)."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import drone_yolo_amd as D
from drone_yolo_amd import hip_ops as H
from oracle import drone_yolo_oracle as O
from tests._util import golden, split_rows

pytestmark = pytest.mark.gpu
X2 = H.F16X2


def up(t, dev, ld=None, c_off=0):
    """CPU NCHW fp32 -> device split-float16 NHWC view (optionally a channel slice, at a multiple of 8, of a wider buffer)."""
    n, c, h, w = t.shape
    c8 = -(-c // 8) * 8
    if ld is None:
        return H.to_nhwc(t.to(dev).contiguous(), X2, c_pad=c8)  # (the 3-channel image comes back as its 8-channel group, zeros behind it)
    buf = torch.zeros((n, h, w, ld), dtype=torch.float32, device=dev).view(X2).permute(0, 3, 1, 2)  # (zero bytes = zero pairs)
    out = buf[:, c_off : c_off + c8]
    H.to_nhwc(t.to(dev).contiguous(), X2, c_pad=c8, out=out)
    return out


def down(t):
    return H.to_nchw_f32(t).cpu()


def test_split_round_trip_keeps_22_bits(device):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 24, 9, 7, generator=g) * torch.logspace(-6, 3, 24).view(1, 24, 1, 1)
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 6.0e-5, -3.0e-8])
    y = down(up(x, device))[:, :24]
    err = (y - x).abs()
    assert float((err / (x.abs() + 1e-30)).where(x.abs() > 1e-3, torch.zeros(())).max()) <= 2.0 ** -21
    assert float(err.where(x.abs() <= 1e-3, torch.zeros(())).max()) <= 2.0 ** -24  # small values: absolute, from the scaled lo half


CASES = [
    # cin, cout, k, s, B, H, W, act, tag
    (8, 32, 3, 2, 2, 64, 64, True, "image layer, Cin 3 padded to 8"),
    (32, 64, 3, 2, 2, 40, 40, True, "repvgg-like s2"),
    (64, 64, 3, 1, 2, 40, 40, True, "3x3 s1 64->64"),
    (96, 64, 1, 1, 2, 40, 40, True, "1x1 K=96"),
    (768, 512, 1, 1, 2, 20, 20, True, "1x1 wide"),
    (256, 256, 3, 1, 1, 20, 20, True, "3x3 deep"),
    (512, 64, 3, 1, 1, 20, 20, True, "3x3 Cin 512 (largest tap table of scale s)"),
    (64, 64, 3, 1, 1, 13, 17, True, "odd spatial (M tail)"),
    (64, 64, 3, 1, 4, 160, 160, True, "large M"),
    (16, 24, 3, 1, 1, 12, 12, True, "n-scale cout 24"),
    (24, 48, 1, 1, 1, 12, 12, False, "cin 24 no act"),
    (128, 128, 3, 2, 2, 40, 40, True, "s2 deep"),
    (160, 160, 3, 1, 1, 24, 24, True, "x-scale 160 (tile 160)"),
    (80, 80, 3, 1, 1, 24, 24, True, "x-scale 80 (tile 80)"),
    # r05: the register-weight kernel of the type (conv3x3_hsplit.hip): cin 32 / 64, cout % 32 == 0, stride 1
    # (maps of at least 3,200 pixels: smaller ones stay on the flat-K kernel)
    (32, 32, 3, 1, 2, 64, 56, True, "hsplit 32->32 (two cout fragments, waves split the rows)"),
    (32, 32, 3, 1, 3, 61, 53, False, "hsplit 32->32 ragged tiles, no act"),
    (32, 64, 3, 1, 1, 64, 64, True, "hsplit 32->64"),
    (64, 128, 3, 1, 2, 57, 67, True, "hsplit 64->128 ragged (two cout groups)"),
    (64, 32, 3, 1, 1, 56, 64, True, "hsplit 64->32"),
    (64, 64, 3, 1, 16, 80, 80, True, "hsplit 64->64 persistent (more tiles than workgroups)"),
]


def _case(case, g, scale_w=1.0):
    cin, cout, k, s, b, h, w, act, tag = case
    creal = 3 if tag.startswith("image") else cin
    x = torch.randn(b, creal, h, w, generator=g)
    wt = torch.randn(cout, creal, k, k, generator=g) * (2.0 / (creal * k * k)) ** 0.5 * scale_w
    bias = torch.randn(cout, generator=g) * 0.2
    return x, wt, bias


@pytest.mark.parametrize("case", CASES, ids=[c[-1] for c in CASES])
def test_split_conv_matches_float64(case, device):
    cin, cout, k, s, b, h, w, act, tag = case
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % 1000)
    x, wt, bias = _case(case, g)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), s, k // 2)
    ref32 = F.conv2d(x, wt, bias, s, k // 2)
    if act:
        ref, ref32 = F.silu(ref), F.silu(ref32)
    pc = H.PackedConv(wt, bias, s, k // 2, 1, act, X2, device)
    y = H.conv2d(up(x, device), pc)
    torch.cuda.synchronize()
    want_h = k == 3 and s == 1 and cin in (32, 64) and cout % 32 == 0 and h * w >= 3200
    assert (H.last_kernel_name() == "conv3x3_hsplit_kernel" if want_h else H.last_kernel_name().startswith("conv_gemm_fk_kernel<split")), H.last_kernel_name()
    assert tuple(y.shape) == tuple(ref.shape)
    got = down(y).double()
    scale = float(ref.abs().max())
    err, err32 = float((got - ref).abs().max()), float((ref32.double() - ref).abs().max())
    # within a few fp32 ulps of the output scale — and never far worse than the fp32 CPU convolution itself is against float64
    assert err <= 4e-6 * scale and err <= 8 * err32 + 1e-6 * scale, (tag, err, err32, scale)


@pytest.mark.parametrize("cin,cout,b,h,w,act", [(128, 64, 2, 20, 24, True), (64, 32, 2, 17, 19, True), (32, 16, 3, 40, 40, True), (16, 8, 1, 9, 11, False), (32, 32, 2, 12, 12, True),
                                                 (64, 16, 1, 14, 10, True)])
def test_split_grouped_conv_matches_float64(cin, cout, b, h, w, act, device):
    """DWConv of the -sf YAML (conv.py:102-107: 3x3 stride 2, groups = gcd(c1, c2)) on split-float16 storage: dy_conv2d_nhwc's grouped
    DY_F16X2 form (fp32 weights, joined inputs) against a float64 CPU convolution, to the tolerance of the dense split kernels;
    input and output as channel slices of wider buffers (the Concat the layer writes into)."""
    import math

    groups = math.gcd(cin, cout)
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(b, cin, h, w, generator=g) * 2.0
    wt = torch.randn(cout, cin // groups, 3, 3, generator=g) * (2.0 / (9 * cin // groups)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.2
    ref = F.conv2d(x.double(), wt.double(), bias.double(), 2, 1, groups=groups)
    ref32 = F.conv2d(x, wt, bias, 2, 1, groups=groups)
    if act:
        ref, ref32 = F.silu(ref), F.silu(ref32)
    pc = H.PackedConv(wt, bias, 2, 1, groups, act, X2, device)
    xin = up(x, device, ld=cin + 16, c_off=8)
    ho, wo = ref.shape[2:]
    buf = torch.zeros((b, ho, wo, cout + 24), dtype=torch.float32, device=device).view(X2).permute(0, 3, 1, 2)
    y = H.conv2d(xin, pc, out=buf[:, 16 : 16 + cout])
    torch.cuda.synchronize()
    assert H.last_kernel_name() == "conv_smallgroup_split_kernel"
    got = down(y).double()
    scale = float(ref.abs().max())
    err, err32 = float((got - ref).abs().max()), float((ref32.double() - ref).abs().max())
    assert err <= 4e-6 * scale and err <= 8 * err32 + 1e-6 * scale, (err, err32, scale)
    assert float(down(buf[:, :16]).abs().max()) == 0.0 and float(down(buf[:, 16 + cout :]).abs().max()) == 0.0  # the neighbours of the slice are untouched
    with pytest.raises(NotImplementedError):
        H.PackedConv(torch.randn(24, 3, 3, 3), torch.zeros(24), 2, 1, 8, True, X2, device)  # 3 inputs per group: not a built form


@pytest.mark.parametrize("how", ["residual", "out_f32", "out_f32_cout10", "slice_io", "x2_up2x", "tiny_weights", "huge_activations",
                                 "residual_big", "residual_big_c32", "slice_io_big"])  # *_big (r05): maps the register-weight kernel takes (conv3x3_hsplit.hip)
def test_split_conv_call_forms(how, device):
    g = torch.Generator().manual_seed(zlib.crc32(how.encode()) % 1000)
    b, cin, cout, h, w = 2, 64, 64, 24, 20
    k, act, kw = 3, True, {}
    big = "_big" in how
    if big:
        h, w = 70, 52
        if how.endswith("c32"):
            cin = cout = 32
        how = how.split("_big")[0]
    if how in ("out_f32", "out_f32_cout10"):
        k, act, cout = 1, False, (10 if how.endswith("10") else 64)
    if how == "x2_up2x":
        k, cin = 1, 192
    x = torch.randn(b, cin, h, w, generator=g) * (300.0 if how == "huge_activations" else 1.0)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5 * (1e-4 if how == "tiny_weights" else 1.0)
    bias = torch.randn(cout, generator=g) * 0.2
    xin = up(x, device)
    if how == "x2_up2x":  # Upsample + Concat folded into a C2f's first 1x1: channels [0, 128) from the half-resolution map, the rest from x2
        xl, xs = torch.randn(b, 128, h // 2, w // 2, generator=g), torch.randn(b, 64, h, w, generator=g)
        x = torch.cat((F.interpolate(xl, scale_factor=2, mode="nearest"), xs), 1)
        xin, kw = up(xl, device), {"x2": up(xs, device), "up2x": True}
    if how == "slice_io":  # input = a channel slice of a wider buffer, output into one
        xin = up(x, device, ld=160, c_off=32)
        obuf = H.alloc_nhwc(b, 128, h, w, X2, device)
        kw = {"out": obuf[:, 64:128]}
    ref = F.conv2d(x.double(), wt.double(), bias.double(), 1, k // 2)
    if act:
        ref = F.silu(ref)
    if how == "residual":
        r = torch.randn(ref.shape, generator=g)
        ref, kw = ref + r.double(), {"residual": up(r, device)}
    if how.startswith("out_f32"):
        ld, c_off = (76, 64) if cout == 10 else (64, 0)  # Detect's fp32 map: box bins [0, 64), class logits [64, 74), pitch 76
        obuf = torch.zeros((b, h, w, ld), dtype=torch.float32, device=device).permute(0, 3, 1, 2)
        kw = {"out": obuf[:, c_off : c_off + cout], "out_f32": True}
    pc = H.PackedConv(wt, bias, 1, k // 2, 1, act, X2, device, for_out_f32=how.startswith("out_f32"))
    y = H.conv2d(xin, pc, **kw)
    torch.cuda.synchronize()
    assert (H.last_kernel_name() == "conv3x3_hsplit_kernel") == (big and not (how == "residual" and cin == 64)), H.last_kernel_name()  # (64 channels with a residual: flat-K)
    got = (y.float().cpu() if how.startswith("out_f32") else down(y)).double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 4e-6 * scale, (how, float((got - ref).abs().max()), scale)
    if how == "out_f32_cout10":
        assert float(obuf[:, 74:].abs().max()) == 0.0  # the pad columns of the pitch stay untouched


def test_split_sppf_pools_are_exact(device):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 20, 20, generator=g)
    buf = H.alloc_nhwc(2, 128, 20, 20, X2, device)
    H.to_nhwc(x.to(device).contiguous(), X2, out=buf[:, :32])
    H.sppf_maxpool3(buf[:, :32], buf[:, 32:64], buf[:, 64:96], buf[:, 96:], 5)
    torch.cuda.synchronize()
    x0 = down(buf[:, :32])
    y1 = F.max_pool2d(x0, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    for i, ref in enumerate((y1, y2, y3)):
        assert torch.equal(down(buf[:, 32 * (i + 1) : 32 * (i + 2)]), ref)


def _build(tag, g):
    from tests.test_model_gpu import _build as b

    return b(tag, g, None)


@pytest.mark.parametrize("tag", ["n64", "n128", "sf_n64", "v8n320", "s640"])  # sf_n64 (r05): the -sf YAML, its DWConv on the split small-group kernel
def test_split_end_to_end_is_bar_exact(tag, device):
    """The fp32 branch of tests/test_model_gpu.py::test_end_to_end_against_reference_vectors, on split-float16 storage: kept sets, classes
    and order identical to the REAL reference's rows, IoU >= 0.999, raw outputs to fp32 round-off."""
    g = golden("e2e.npz")
    m, d, sd, model, x = _build(tag, g)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="f16x2", device=0))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    y = cf.pred.cpu()
    if f"{tag}__y" in g.files:
        yref = torch.from_numpy(g[f"{tag}__y"])
    else:
        yref, y = torch.from_numpy(g[f"{tag}__y_sub"]), y[:, :, ::37]
    box_err, cls_err = float((y[:, :4] - yref[:, :4]).abs().max()), float((y[:, 4:] - yref[:, 4:]).abs().max())
    assert box_err < 2e-2 and cls_err < 1e-4, (box_err, cls_err)
    counts = cf.nms.count.cpu().tolist()
    assert counts == [int(v) for v in g[f"{tag}__n"]]
    exp_rows = split_rows(g[f"{tag}__det"], g[f"{tag}__n"])
    exp_idx = split_rows(g[f"{tag}__det_idx"], g[f"{tag}__n"])
    for i, c in enumerate(counts):
        got_idx, got_cls = cf.nms.index[i, :c].cpu().numpy(), cf.nms.out[i, :c, 5].cpu().numpy()
        assert sorted(got_idx.tolist()) == sorted(exp_idx[i].tolist()), f"{tag} image {i}: kept anchor sets differ"
        cls_of = {int(a): int(k) for a, k in zip(exp_idx[i], exp_rows[i][:, 5])}
        assert all(cls_of[int(a)] == int(k) for a, k in zip(got_idx, got_cls))
        score_of = {int(a): float(sc) for a, sc in zip(exp_idx[i], exp_rows[i][:, 4])}
        for k in np.nonzero(got_idx != exp_idx[i])[0]:
            assert abs(score_of[int(got_idx[k])] - float(exp_rows[i][k, 4])) < 2e-6, f"{tag} image {i}: order differs at rank {k}"


@pytest.mark.parametrize("tag", ["s640bench", "s640b4", "s640b4lo"])
def test_split_bench_configuration_is_bar_exact(tag, device):
    """BASELINE config 2's fixtures (the REAL reference's fp32 CPU rows, tests/golden/big.npz) on split-float16 storage: the fp32 assertions."""
    from drone_yolo_amd.utils import parity as PR
    from tests.test_model_gpu import _bench_model

    meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", tag)
    model = _bench_model(meta, device)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="f16x2", device=0))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    g = golden("big.npz")
    y_sub = cf.pred[:, :, ::199].cpu()
    box_err = float((y_sub[:, :4] - torch.from_numpy(g[f"{tag}__y_sub"])[:, :4]).abs().max())
    cls_err = float((y_sub[:, 4:] - torch.from_numpy(g[f"{tag}__y_sub"])[:, 4:]).abs().max())
    par = PR.detection_parity(cf.nms, exp_rows, exp_idx, conf=0.25, margin=0.0)
    assert box_err < 2e-2 and cls_err < 1e-4, (box_err, cls_err)
    assert par["counts_equal"] and par["kept_sets_identical"] and par["match_rate"] == 1.0 and par["iou_min"] >= 0.999, par
    names = [fn.__name__ for fn, _, _ in cf.plan.ops]
    assert "dy_detect_head_decode" in names and "dy_stem2_fused" in names  # the type's fused Detect tail (1x1 x 2 + decode + filter) and fused image stem pair ran


@pytest.mark.parametrize("shape", [(2, 3, 64, 64, 32), (1, 3, 50, 70, 16), (3, 3, 640, 640, 32), (1, 1, 32, 36, 64)], ids=["64x64", "odd 50x70 cout 16", "640x640", "cin 1 cout 64"])
def test_split_stem_matches_float64(shape, device):
    """dy_stem_conv3x3s2_nchw with DY_F16X2: fp32 NCHW image -> Conv(cin, cout, 3, 2) + SiLU -> split NHWC, image and weights split on the fly."""
    b, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(h * 7 + cout)
    x = torch.rand(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.2
    ref = F.silu(F.conv2d(x.double(), wt.double(), bias.double(), 2, 1))
    y = H.stem_conv(x.to(device).contiguous(), H.PackedStem(wt, bias, True, X2, device))
    torch.cuda.synchronize()
    assert H.last_kernel_name() == "conv_stem_split_kernel" and tuple(y.shape) == tuple(ref.shape)
    err = float((down(y).double() - ref).abs().max())
    assert err <= 4e-6 * float(ref.abs().max()), err


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 52, 76), (3, 640, 640), (2, 4, 20)], ids=["64x64", "ragged 52x76", "640x640", "one partial tile"])
def test_split_fused_stem_pair_matches_float64_and_the_layer_path(shape, device):
    """dy_stem2_fused with DY_F16X2 (r05): fp32 image -> Conv(3, 32, 3, 2) + SiLU -> 3x3 stride-2 32 -> 64 + SiLU in one kernel, the half-resolution map
    kept in LDS as (hi, lo) halves — against a float64 CPU chain to the dense split kernels' tolerance, and against the two launches it replaces."""
    b, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    x = torch.rand(b, 3, h, w, generator=g)
    w0, b0 = torch.randn(32, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5, torch.randn(32, generator=g) * 0.2
    w1, b1 = torch.randn(64, 32, 3, 3, generator=g) * (2.0 / 288) ** 0.5, torch.randn(64, generator=g) * 0.2
    ref = F.silu(F.conv2d(F.silu(F.conv2d(x.double(), w0.double(), b0.double(), 2, 1)), w1.double(), b1.double(), 2, 1))
    assert H.stem2_fused_supported(3, 32, 64, h, w, X2)
    xd = x.to(device).contiguous()
    y = H.stem2_fused(xd, H.PackedStem2(w0, b0, True, w1, b1, True, X2, device))
    torch.cuda.synchronize()
    assert H.last_kernel_name() == "stem2_split_kernel" and tuple(y.shape) == tuple(ref.shape)
    got = down(y).double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 6e-6 * scale, float((got - ref).abs().max()) / scale
    t = H.stem_conv(xd, H.PackedStem(w0, b0, True, X2, device))
    y2 = H.conv2d(t, H.PackedConv(w1, b1, 2, 1, 1, True, X2, device))
    torch.cuda.synchronize()
    assert float((down(y2).double() - got).abs().max()) <= 2e-6 * scale  # same roundings of the intermediate, another summation order in layer 1
    # into a channel slice of a wider buffer (a Concat the layer writes into): the neighbours stay untouched
    buf = torch.zeros((b, h // 4, w // 4, 64 + 24), dtype=torch.float32, device=device).view(X2).permute(0, 3, 1, 2)
    H.stem2_fused(xd, H.PackedStem2(w0, b0, True, w1, b1, True, X2, device), out=buf[:, 8:72])
    torch.cuda.synchronize()
    assert torch.equal(down(buf[:, 8:72]), down(y)) and float(down(buf[:, :8]).abs().max()) == 0.0 and float(down(buf[:, 72:]).abs().max()) == 0.0


def test_split_full_size_properties(device):
    """Drone-YOLO-s 640x640 on split-float16 storage, size-independent properties at a real batch (no oracle at this size): images are independent (a batch
    equals its images run one by one, bit for bit), a permuted batch gives permuted outputs, the hipGraph replay equals the recorded pass, and the NMS
    output invariants hold (counts <= max_det, scores sorted and above conf, boxes inside the image, kept anchors unique)."""
    g = golden("e2e.npz")
    m, d, sd, model, _ = _build("s640", g)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype="f16x2", device=0, graph=True))
    x = torch.rand(8, 3, 640, 640, generator=torch.Generator().manual_seed(5)).to(device)
    cf = pred.forward_device(x)
    torch.cuda.synchronize()
    out, cnt, idx, y = cf.nms.out.clone(), cf.nms.count.clone(), cf.nms.index.clone(), cf.pred.clone()
    cf2 = pred.forward_device(x.clone())  # the captured graph, another input buffer with the same contents
    torch.cuda.synchronize()
    assert cf2 is cf and cf.graph is not None and torch.equal(cf.pred, y) and torch.equal(cf.nms.out, out) and torch.equal(cf.nms.count, cnt)
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=device)
    cf = pred.forward_device(x[perm].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(cf.pred, y[perm]) and torch.equal(cf.nms.out, out[perm]) and torch.equal(cf.nms.count, cnt[perm])
    single = pred.forward_device(x[2:3].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(single.pred[0], y[2]) and torch.equal(single.nms.out[0], out[2])
    for i in range(8):
        c = int(cnt[i])
        assert 0 < c <= 300
        sc = out[i, :c, 4]
        assert bool((sc[:-1] >= sc[1:]).all()) and float(sc.min()) > 0.25
        assert float(out[i, :c, :4].min()) >= 0 and float(out[i, :c, :4].max()) <= 640
        assert len(set(idx[i, :c].tolist())) == c


def test_split_batch_of_256_equals_batches_of_8(device):
    """BASELINE config 2's batch (256 images of 640 x 640, bench.py's weights) on split-float16 storage: the P2 maps of the batch are 3.4 GB views, the
    reach of the register-weight kernel's 32-bit buffer offsets (conv3x3_hsplit.hip) — images at the start, the middle and the end of the batch come out
    bit for bit as in a batch of 8."""
    import bench

    model = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False)
    model.load_state_dict(bench.synthetic_state_dict(model, seed=0))
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, device=0, graph=True))
    assert pred.dtype == X2
    x = torch.rand(256, 3, 640, 640, generator=torch.Generator().manual_seed(7)).to(device)
    cf = pred.forward_device(x)
    torch.cuda.synchronize()
    y, out, cnt = cf.pred.clone(), cf.nms.out.clone(), cf.nms.count.clone()
    for lo in (0, 120, 248):
        cs = pred.forward_device(x[lo : lo + 8].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(cs.pred, y[lo : lo + 8]) and torch.equal(cs.nms.out, out[lo : lo + 8]) and torch.equal(cs.nms.count, cnt[lo : lo + 8]), lo
