"""ORACLE — test infrastructure, not product code.

A CPU restatement (plain PyTorch fp32, NCHW, functional) of the reference's detection hot path,
written from the reference sources and cited line by line.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package (``drone-yolo_amd/``) never does.

Pinning: ``oracle/make_golden.py`` imports the real reference from ``/root/reference`` in the build
container and checks this restatement against it layer by layer and end to end; the vectors it
writes live in ``tests/golden/`` (the reference cannot travel to the GPU box).  The reference ships
no tests or golden vectors of its own, and ``torchvision.ops.nms`` — the NMS the
reference calls at utils/ops.py:312 — is absent from the container (torchvision>=0.9.0, unpinned,
requirements.txt:14), so **parity is unpinned at the NMS boundary**: ``nms_greedy`` below restates
torchvision's published CPU algorithm (stable descending sort, suppress iff IoU > thr) and is pinned
only by brute-force property tests.

Everything takes a ``state dict`` with the reference's key names (``model.{i}.conv.weight`` ...).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # utils/torch_utils.py:423-433 (initialize_weights sets eps=1e-3 on every BatchNorm2d)
Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ---- YAML walk (nn/tasks.py:929-1090) ------------------------------------------------------------------


def make_divisible(x, divisor):
    """utils/ops.py:130-143."""
    return math.ceil(x / divisor) * divisor


def resolve_layers(d: dict, ch: int = 3):
    """[(index, from, module name, resolved args)] with channel/depth scaling applied — the arithmetic of
    parse_model (tasks.py:1012-1035, 1052-1063) with RepVGGBlock treated as a base module."""
    nc = d["nc"]
    depth, width, max_ch = d["scales"][d["scale"]] if d.get("scales") else (d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0), float("inf"))
    chs: List[int] = []
    out = []
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        args = [nc if a == "nc" else a for a in args]
        n = max(round(n * depth), 1) if n > 1 else n
        c_in = ch if i == 0 else (chs[f] if isinstance(f, int) else None)
        if m in ("Conv", "RepVGGBlock", "C2f", "SPPF", "DWConv"):
            c2 = args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_ch) * width, 8)
            args = [c_in, c2, *args[1:]]
            if m == "C2f":
                args.insert(2, n)
        elif m == "Concat":
            c2 = sum(chs[x] for x in f)
        elif m == "Detect":
            args = [args[0], [chs[x] for x in f]]
            c2 = None
        else:  # nn.Upsample
            c2 = chs[f]
        out.append((i, f, m, args))
        chs.append(c2)
    return out


# ---- operators ----------------------------------------------------------------------------------------------


def autopad(k, p=None):
    """nn/modules/conv.py:28-34."""
    return k // 2 if p is None else p


def bn_eval(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


BN_MOMENTUM = 0.03  # utils/torch_utils.py:423-433 (initialize_weights)


def bn_apply(x: Tensor, sd: SD, p: str, fused) -> Tensor:
    """BatchNorm2d of the unfused module graph: eval statistics, or — ``fused == "train"`` — batch statistics with the
    running buffers of ``sd`` updated in place (module.train(), the form engine/trainer.py:381 runs)."""
    if fused == "train":
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], True, BN_MOMENTUM, BN_EPS)
    return bn_eval(x, sd, p)


def fuse_conv_bn(w: Tensor, sd: SD, bn: str) -> Tuple[Tensor, Tensor]:
    """utils/torch_utils.py:242-269: W' = diag(g/sqrt(eps+var)) W ; b' = beta - g*mean/sqrt(var+eps)."""
    w_bn = torch.diag(sd[bn + ".weight"].div(torch.sqrt(BN_EPS + sd[bn + ".running_var"])))
    wf = torch.mm(w_bn, w.view(w.shape[0], -1)).view(w.shape)
    bf = sd[bn + ".bias"] - sd[bn + ".weight"].mul(sd[bn + ".running_mean"]).div(torch.sqrt(sd[bn + ".running_var"] + BN_EPS))
    return wf, bf


def conv_block(x: Tensor, sd: SD, p: str, k: int, s: int, g: int = 1, fused: bool = True) -> Tensor:
    """Conv.forward / forward_fuse (conv.py:49-55): SiLU(BN(conv(x))) or SiLU(conv'(x) + b')."""
    w = sd[p + ".conv.weight"]
    if fused is True:
        wf, bf = fuse_conv_bn(w, sd, p + ".bn")
        return F.silu(F.conv2d(x, wf, bf, s, autopad(k), 1, g))
    return F.silu(bn_apply(F.conv2d(x, w, None, s, autopad(k), 1, g), sd, p + ".bn", fused))


def repvgg_block(x: Tensor, sd: SD, p: str, stride: int, has_identity: bool, fused=False) -> Tensor:
    """RepVGGBlock.forward, non-deploy (block.py:1480-1490): SiLU(BN(conv3x3) + BN(conv1x1) + BN(x)).
    BaseModel.fuse() does not touch this block (tasks.py:193-221), so this is also the predict form."""
    dense = bn_apply(F.conv2d(x, sd[p + ".rbr_dense.conv.weight"], None, stride, 1), sd, p + ".rbr_dense.bn", fused)
    one = bn_apply(F.conv2d(x, sd[p + ".rbr_1x1.conv.weight"], None, stride, 0), sd, p + ".rbr_1x1.bn", fused)
    idt = bn_apply(x, sd, p + ".rbr_identity", fused) if has_identity else 0
    return F.silu(dense + one + idt)


def repvgg_equivalent(sd: SD, p: str, has_identity: bool, in_channels: int, groups: int = 1) -> Tuple[Tensor, Tensor]:
    """get_equivalent_kernel_bias (block.py:1446-1478): one 3x3 kernel + bias for the three branches."""

    def fuse(kernel, bn):
        std = (sd[bn + ".running_var"] + BN_EPS).sqrt()
        t = (sd[bn + ".weight"] / std).reshape(-1, 1, 1, 1)
        return kernel * t, sd[bn + ".bias"] - sd[bn + ".running_mean"] * sd[bn + ".weight"] / std

    k3, b3 = fuse(sd[p + ".rbr_dense.conv.weight"], p + ".rbr_dense.bn")
    k1, b1 = fuse(sd[p + ".rbr_1x1.conv.weight"], p + ".rbr_1x1.bn")
    k, b = k3 + F.pad(k1, [1, 1, 1, 1]), b3 + b1
    if has_identity:
        input_dim = in_channels // groups
        kid = torch.zeros((in_channels, input_dim, 3, 3))
        for i in range(in_channels):
            kid[i, i % input_dim, 1, 1] = 1
        ki, bi = fuse(kid, p + ".rbr_identity")
        k, b = k + ki, b + bi
    return k, b


def bottleneck(x: Tensor, sd: SD, p: str, add: bool, fused: bool) -> Tensor:
    """Bottleneck.forward (block.py:348-350), k=(3,3), e=1.0 inside C2f."""
    y = conv_block(conv_block(x, sd, p + ".cv1", 3, 1, fused=fused), sd, p + ".cv2", 3, 1, fused=fused)
    return x + y if add else y


def c2f(x: Tensor, sd: SD, p: str, n: int, shortcut: bool, fused: bool) -> Tensor:
    """C2f.forward (block.py:237-242)."""
    y = list(conv_block(x, sd, p + ".cv1", 1, 1, fused=fused).chunk(2, 1))
    for i in range(n):
        y.append(bottleneck(y[-1], sd, f"{p}.m.{i}", shortcut, fused))
    return conv_block(torch.cat(y, 1), sd, p + ".cv2", 1, 1, fused=fused)


def sppf(x: Tensor, sd: SD, p: str, k: int, fused: bool) -> Tensor:
    """SPPF.forward (block.py:187-191)."""
    y = [conv_block(x, sd, p + ".cv1", 1, 1, fused=fused)]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], k, 1, k // 2))
    return conv_block(torch.cat(y, 1), sd, p + ".cv2", 1, 1, fused=fused)


def make_anchors(shapes: Sequence[Tuple[int, int]], strides: Sequence[float], offset: float = 0.5):
    """utils/tal.py:333-345."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(end=w, dtype=torch.float32) + offset
        sy = torch.arange(end=h, dtype=torch.float32) + offset
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=torch.float32))
    return torch.cat(pts), torch.cat(st)


def dfl(x: Tensor, reg_max: int = 16) -> Tensor:
    """DFL.forward (block.py:73-76): softmax over the bins, then the frozen arange 1x1 conv."""
    b, _, a = x.shape
    w = torch.arange(reg_max, dtype=torch.float32).view(1, reg_max, 1, 1)
    return F.conv2d(x.view(b, 4, reg_max, a).transpose(2, 1).softmax(1), w).view(b, 4, a)


def dist2bbox(distance: Tensor, anchor_points: Tensor, xywh: bool = True, dim: int = -1) -> Tensor:
    """utils/tal.py:348-357."""
    lt, rb = distance.chunk(2, dim)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


def detect_head(feats: List[Tensor], sd: SD, p: str, nc: int, fused: bool) -> List[Tensor]:
    """Detect.forward up to the training return (head.py:64-72), legacy=True branches (head.py:43-47)."""
    out = []
    for i, x in enumerate(feats):
        def branch(name):
            y = conv_block(x, sd, f"{p}.{name}.{i}.0", 3, 1, fused=fused)
            y = conv_block(y, sd, f"{p}.{name}.{i}.1", 3, 1, fused=fused)
            return F.conv2d(y, sd[f"{p}.{name}.{i}.2.weight"], sd[f"{p}.{name}.{i}.2.bias"])
        out.append(torch.cat((branch("cv2"), branch("cv3")), 1))
    return out


def detect_decode(x: List[Tensor], strides: Sequence[float], nc: int, reg_max: int = 16) -> Tensor:
    """Detect._inference (head.py:100-131) -> (B, 4+nc, A)."""
    b = x[0].shape[0]
    no = nc + 4 * reg_max
    x_cat = torch.cat([xi.view(b, no, -1) for xi in x], 2)
    anchors, st = (t.transpose(0, 1) for t in make_anchors([xi.shape[2:] for xi in x], strides, 0.5))
    box, cls = x_cat.split((reg_max * 4, nc), 1)
    dbox = dist2bbox(dfl(box, reg_max), anchors.unsqueeze(0), xywh=True, dim=1) * st
    return torch.cat((dbox, cls.sigmoid()), 1)


# ---- whole model --------------------------------------------------------------------------------------------


def model_strides(layers) -> List[float]:
    """Strides of the Detect inputs = what the zeros(1,ch,256,256) probe of tasks.py:324-337 measures."""
    cum: List[float] = []
    for i, f, m, args in layers:
        f0 = f if isinstance(f, int) else f[0]
        s_in = 1.0 if i == 0 else cum[f0 if f0 >= 0 else i + f0]
        if m in ("Conv", "RepVGGBlock", "DWConv"):
            cum.append(s_in * (args[3] if len(args) > 3 else 1))
        elif m == "nn.Upsample":
            cum.append(s_in / 2)
        else:
            cum.append(s_in)
    det = layers[-1]
    return [cum[j] for j in det[1]]


def forward(d: dict, sd: SD, x: Tensor, fused: bool = True, return_all: bool = False):
    """BaseModel._predict_once (tasks.py:134-161) over the YAML graph, eval mode.

    fused=True is the predictor's form (AutoBackend fuse=True, autobackend.py:143-155: Conv+BN folded,
    RepVGGBlock left 3-branch); fused=False is the freshly built module graph; fused="train" is that graph in
    training mode (batch-statistics BatchNorm updating the running buffers in ``sd``, Detect returning the raw maps).
    Returns (y, feats): decoded (B, 4+nc, A) and the raw per-level head outputs, like Detect eval.
    """
    layers = resolve_layers(d, x.shape[1])
    strides = model_strides(layers)
    ys: List[Optional[Tensor]] = []
    cur = x
    for i, f, m, args in layers:
        if f != -1:
            cur = ys[f] if isinstance(f, int) else [cur if j == -1 else ys[j] for j in f]
        p = f"model.{i}"
        if m == "Conv":
            cur = conv_block(cur, sd, p, args[2], args[3], fused=fused)
        elif m == "DWConv":
            cur = conv_block(cur, sd, p, args[2], args[3], g=math.gcd(args[0], args[1]), fused=fused)
        elif m == "RepVGGBlock":
            cur = repvgg_block(cur, sd, p, args[3], has_identity=(args[0] == args[1] and args[3] == 1), fused=fused)
        elif m == "C2f":
            cur = c2f(cur, sd, p, args[2], bool(args[3]) if len(args) > 3 else False, fused)
        elif m == "SPPF":
            cur = sppf(cur, sd, p, args[2], fused)
        elif m == "nn.Upsample":
            cur = F.interpolate(cur, scale_factor=2.0, mode="nearest")
        elif m == "Concat":
            cur = torch.cat(cur, 1)
        elif m == "Detect":
            feats = detect_head(cur, sd, p, args[0], fused)
            cur = feats if fused == "train" else (detect_decode(feats, strides, args[0]), feats)  # head.py:71-74
        else:
            raise NotImplementedError(m)
        ys.append(cur)
    return (cur, ys) if return_all else cur


# ---- post-processing ------------------------------------------------------------------------------------------


def xywh2xyxy(x: Tensor) -> Tensor:
    """utils/ops.py:432-449 (output always fp32)."""
    y = torch.empty_like(x, dtype=torch.float32)
    xy = x[..., :2]
    wh = x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def nms_greedy(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision.ops.nms CPU semantics (call site utils/ops.py:312; library absent, see module docstring):
    stable descending sort of scores; keep i unless suppressed; suppress j>i iff
    inter/(area_i + area_j - inter) > iou_threshold, all in fp32, no epsilon. Returns kept indices."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    with np.errstate(divide="ignore", invalid="ignore"):
        for _i in range(n):
            i = order[_i]
            if suppressed[i]:
                continue
            keep.append(i)
            rest = order[_i + 1 :]
            xx1 = np.maximum(x1[i], x1[rest])
            yy1 = np.maximum(y1[i], y1[rest])
            xx2 = np.minimum(x2[i], x2[rest])
            yy2 = np.minimum(y2[i], y2[rest])
            w = np.maximum(np.float32(0), xx2 - xx1)
            h = np.maximum(np.float32(0), yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(prediction: Tensor, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, max_det=300,
                        nc=0, max_nms=30000, max_wh=7680, return_index=False, multi_label=False):
    """utils/ops.py:181-332 without the wall-clock break (ops.py:328-330): the single-label path (best class per anchor, ops.py:290-291)
    and the validator's ``multi_label`` path (one candidate per (anchor, class) pair scored above conf, ops.py:286-288).
    ``return_index``: also the anchor index of every kept row."""
    prediction = prediction.clone()
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    nm = prediction.shape[1] - nc - 4
    mi = 4 + nc
    xc = prediction[:, 4:mi].amax(1) > conf_thres  # ops.py:250
    multi_label &= nc > 1  # ops.py:255
    prediction = prediction.transpose(-1, -2)  # ops.py:257
    prediction[..., :4] = xywh2xyxy(prediction[..., :4])  # ops.py:259-260
    output = [torch.zeros((0, 6 + nm))] * bs
    indices = [torch.zeros((0,), dtype=torch.int64)] * bs
    for xi, x in enumerate(prediction):
        aidx = torch.nonzero(xc[xi]).flatten()
        x = x[xc[xi]]  # ops.py:269
        if not x.shape[0]:
            continue
        box, cls, mask = x.split((4, nc, nm), 1)
        if multi_label:  # ops.py:286-288
            i, j = torch.where(cls > conf_thres)
            x = torch.cat((box[i], x[i, 4 + j, None], j[:, None].float(), mask[i]), 1)
            aidx = aidx[i]
        else:
            conf, j = cls.max(1, keepdim=True)  # ops.py:290
            sel = conf.view(-1) > conf_thres
            x = torch.cat((box, conf, j.float(), mask), 1)[sel]  # ops.py:291
            aidx = aidx[sel]
        if classes is not None:
            sel = (x[:, 5:6] == torch.tensor(classes)).any(1)  # ops.py:294-295
            x, aidx = x[sel], aidx[sel]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:  # ops.py:301-302
            o = x[:, 4].argsort(descending=True)[:max_nms]
            x, aidx = x[o], aidx[o]
        c = x[:, 5:6] * (0 if agnostic else max_wh)  # ops.py:305
        boxes, scores = x[:, :4] + c, x[:, 4]  # ops.py:311
        i = torch.from_numpy(nms_greedy(boxes.numpy(), scores.numpy(), iou_thres))  # ops.py:312
        i = i[:max_det]  # ops.py:313
        output[xi] = x[i]
        indices[xi] = aidx[i]
    return (output, indices) if return_index else output


def clip_boxes(boxes: Tensor, shape) -> Tensor:
    """utils/ops.py:335-354."""
    boxes[..., 0] = boxes[..., 0].clamp(0, shape[1])
    boxes[..., 1] = boxes[..., 1].clamp(0, shape[0])
    boxes[..., 2] = boxes[..., 2].clamp(0, shape[1])
    boxes[..., 3] = boxes[..., 3].clamp(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes: Tensor, img0_shape, ratio_pad=None, padding=True) -> Tensor:
    """utils/ops.py:92-127 (xyxy)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    if padding:
        boxes[..., 0] -= pad[0]
        boxes[..., 1] -= pad[1]
        boxes[..., 2] -= pad[0]
        boxes[..., 3] -= pad[1]
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


def predict(d: dict, sd: SD, im: Tensor, conf=0.25, iou=0.7, max_det=300, classes=None, agnostic=False, fused=True):
    """model.predict on a tensor source (engine/predictor.py:221-306 + detect/predict.py:23-73): list of (n_i, 6)."""
    y, _ = forward(d, sd, im.float(), fused=fused)
    preds = non_max_suppression(y, conf, iou, classes, agnostic, max_det=max_det, nc=d["nc"])
    return [torch.cat((scale_boxes(im.shape[2:], p[:, :4].clone(), im.shape[2:]), p[:, 4:]), 1) if len(p) else p for p in preds]


# ---- test-time augmentation (predict(augment=True)) -------------------------------------------------------------


def scale_img(img: Tensor, ratio: float = 1.0, gs: int = 32) -> Tensor:
    """utils/torch_utils.py:436-445: bilinear resize to (int(h r), int(w r)), then padded right / bottom with 0.447 up to the next multiples of ``gs`` of (h r, w r)."""
    if ratio == 1.0:
        return img
    h, w = img.shape[2:]
    s = (int(h * ratio), int(w * ratio))
    img = F.interpolate(img, size=s, mode="bilinear", align_corners=False)
    h2, w2 = (math.ceil(v * ratio / gs) * gs for v in (h, w))
    return F.pad(img, [0, w2 - s[1], 0, h2 - s[0]], value=0.447)


def predict_augment(d: dict, sd: SD, x: Tensor, fused: bool = True) -> Tensor:
    """DetectionModel._predict_augment (nn/tasks.py:347-383): the image at scales 1, 0.83 (mirrored left-right) and 0.67, every pass's boxes scaled back
    (and mirrored back) into the input's frame (_descale_pred), the coarsest level of the first pass and the finest level of the last one dropped
    (_clip_augmented), all anchors side by side: (B, 4 + nc, A_total)."""
    img_size = x.shape[-2:]
    layers = resolve_layers(d, x.shape[1])
    gs = int(max(model_strides(layers)))
    ys = []
    for si, fi in zip((1, 0.83, 0.67), (None, 3, None)):
        xi = scale_img(x.flip(fi) if fi else x, si, gs=gs)
        yi = forward(d, sd, xi, fused=fused)[0].clone()
        yi[:, :4] /= si  # de-scale
        if fi == 3:
            yi[:, 0] = img_size[1] - yi[:, 0]  # de-flip lr
        ys.append(yi)
    nl = len(model_strides(layers))  # detection levels
    g = sum(4 ** k for k in range(nl))
    i = (ys[0].shape[-1] // g) * 1
    ys[0] = ys[0][..., :-i]  # large: without its coarsest level
    i = (ys[-1].shape[-1] // g) * 4 ** (nl - 1)
    ys[-1] = ys[-1][..., i:]  # small: without its finest level
    return torch.cat(ys, -1)


# ---- deterministic weights shared by both sides ---------------------------------------------------------------


def seeded_state_dict(template: Dict[str, Tensor], seed: int, cls_bias: Optional[float] = None) -> SD:
    """Fill a state dict (keys + shapes from ``template``) from one CPU generator in sorted-key order, so
    the reference import, this oracle and the device package get bit-identical weights from a seed
    (fixtures then hold outputs only).  Conv weights ~ N(0, 2/fan_in) (keeps activations O(1) through the whole graph), BN affine near identity with
    non-trivial running statistics; ``dfl.conv.weight`` keeps its arange; ``cls_bias`` (if given) replaces
    the bias of the last conv of every Detect class branch (head.py:133-144 initialises it to ~-11,
    which leaves no candidate above conf)."""
    g = torch.Generator().manual_seed(seed)
    out: SD = {}
    for k in sorted(template.keys()):
        t = template[k]
        shape = tuple(t.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=t.dtype)
        elif ".dfl." in k or k.startswith("dfl."):
            out[k] = torch.arange(shape[1], dtype=torch.float32).view(shape)
        elif k.endswith("running_mean"):
            out[k] = torch.randn(shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            out[k] = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif k.endswith("weight") and len(shape) == 1:  # BatchNorm affine scale
            out[k] = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif k.endswith("bias"):
            out[k] = torch.randn(shape, generator=g) * 0.1
        else:  # conv weight (cout, cin/g, k, k)
            fan_in = shape[1] * shape[2] * shape[3]
            out[k] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
    if cls_bias is not None:
        for k in out:
            if ".cv3." in k and k.endswith(".2.bias"):
                out[k] = out[k] * 0 + cls_bias + torch.linspace(-0.3, 0.3, out[k].numel())
    return out
