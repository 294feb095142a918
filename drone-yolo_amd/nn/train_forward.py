"""Training-mode forward of the Drone-YOLO modules (module.train()): what BaseModel._predict_once does in the
reference (nn/tasks.py:134-161) with every module in training mode, built from the autograd ops in autograd_ops.py.

Dispatch is by module type; parameters stay in the reference-compatible children (``conv.weight``, ``bn.weight`` ...).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

import os

from .. import hip_ops as H
from . import autograd_ops as A

_STEM_U8 = os.environ.get("DYOLO_STEM_U8", "1") != "0"  # training: layer 0 straight from the uint8 batch


def conv_train(m, x, need_dx: bool = True, img_u8=None):
    c = m.conv
    if c.groups != 1:  # DWConv of the -sf YAML
        return A.GroupedConvBnAct.apply(x, c.weight, m.bn.weight, m.bn.bias, m.bn, c.stride[0], c.padding[0], c.groups, isinstance(m.act, nn.SiLU))
    stem = None
    if (img_u8 is not None and _STEM_U8 and c.kernel_size[0] == 3 and c.stride[0] == 2 and c.padding[0] == 1 and c.weight.shape[1] <= 3
            and c.weight.shape[0] % 16 == 0 and c.weight.shape[0] <= 80 and x.dtype in (torch.bfloat16, torch.float16) and c.weight.is_cuda):
        stem = img_u8  # the first layer reads the uint8 batch itself (dy_stem_conv3x3s2_nchw_u8)
    return A.ConvBnAct.apply(x, c.weight, m.bn.weight, m.bn.bias, m.bn, c.stride[0], c.padding[0], isinstance(m.act, nn.SiLU), need_dx, stem)


def repvgg_train(m, x):
    if getattr(m, "rbr_identity", None) is not None or hasattr(m, "rbr_reparam") or m.groups != 1:
        raise NotImplementedError("RepVGGBlock training is built for the stride-2 two-branch form of the Drone-YOLO YAMLs")
    d, o = m.rbr_dense, m.rbr_1x1
    return A.RepVGGTrain.apply(x, d.conv.weight, d.bn.weight, d.bn.bias, o.conv.weight, o.bn.weight, o.bn.bias, d.bn, o.bn, m.stride)


def conv_pair_train(m1, m2, x):
    """m2(m1(x)) for two plain Conv modules whose intermediate has no other consumer: one autograd node (A.ConvBnAct2)."""
    c1, c2 = m1.conv, m2.conv
    if c1.groups != 1 or c2.groups != 1 or x.shape[1] != c1.weight.shape[1]:
        return conv_train(m2, conv_train(m1, x))
    geo = ((c1.stride[0], c1.padding[0], isinstance(m1.act, nn.SiLU)), (c2.stride[0], c2.padding[0], isinstance(m2.act, nn.SiLU)))
    return A.ConvBnAct2.apply(x, c1.weight, m1.bn.weight, m1.bn.bias, c2.weight, m2.bn.weight, m2.bn.bias, m1.bn, m2.bn, geo, True)


def bottleneck_train(m, x):
    y = conv_pair_train(m.cv1, m.cv2, x)
    return A.AddT.apply(x, y) if m.add else y


def c2f_train(m, x):
    """One autograd node per block (A.C2fTrain) for the plain form of the Drone-YOLO YAMLs; ``m.fuse_block_train = False`` runs
    the op-by-op graph (chunk, Bottlenecks, cat as separate nodes)."""
    plain = all(cv.conv.groups == 1 for mm in m.m for cv in (mm.cv1, mm.cv2)) and m.cv1.conv.groups == 1 and m.cv2.conv.groups == 1
    if plain and getattr(m, "fuse_block_train", True):
        convs = [m.cv1] + [cv for mm in m.m for cv in (mm.cv1, mm.cv2)] + [m.cv2]
        if all(isinstance(cv.act, nn.SiLU) for cv in convs):
            return A.C2fTrain.apply(x, m, *[t for cv in convs for t in (cv.conv.weight, cv.bn.weight, cv.bn.bias)])
    a, b = A.Chunk2.apply(conv_train(m.cv1, x))
    ys = [a, b]
    for mm in m.m:
        ys.append(bottleneck_train(mm, ys[-1]))
    return conv_train(m.cv2, A.ConcatC.apply(*ys))


def sppf_train(m, x):
    return conv_train(m.cv2, A.SppfPool.apply(conv_train(m.cv1, x), m.k))


def detect_train(m, xs: List[torch.Tensor]) -> List[torch.Tensor]:
    """Detect.forward's training return (head.py:64-72): one (N, 4*reg_max+nc, H, W) fp32 map per level."""
    if not m.legacy:
        raise NotImplementedError("Detect training is built for the legacy (two 3x3) class branch of the v8 YAMLs")
    out = []
    for i, x in enumerate(xs):
        b, c = m.cv2[i], m.cv3[i]
        xb = conv_pair_train(b[0], b[1], x)
        xc = conv_pair_train(c[0], c[1], x)
        out.append(A.HeadTail.apply(xb, xc, b[2].weight, b[2].bias, c[2].weight, c[2].bias))
    return out


def module_train(m, x, first: bool = False, img_u8=None):
    from .modules import C2f, Concat, Conv, Detect, RepVGGBlock, SPPF, Upsample

    if isinstance(m, nn.Sequential):
        for mm in m:
            x = module_train(mm, x)
        return x
    if isinstance(m, Conv):
        return conv_train(m, x, need_dx=not first, img_u8=img_u8 if first else None)
    if isinstance(m, RepVGGBlock):
        return repvgg_train(m, x)
    if isinstance(m, C2f):
        return c2f_train(m, x)
    if isinstance(m, SPPF):
        return sppf_train(m, x)
    if isinstance(m, Upsample):
        return A.Upsample2x.apply(x)
    if isinstance(m, Concat):
        return A.ConcatC.apply(*x)
    if isinstance(m, Detect):
        return detect_train(m, x)
    raise NotImplementedError(f"no training forward for {type(m).__name__}")


def model_train_forward(model, img: torch.Tensor, dtype: torch.dtype):
    """img: (N, 3, H, W) uint8 or float on the device -> Detect's per-level training outputs."""
    H.require_device(img, "training image")
    img_u8 = None
    if img.dtype == torch.uint8:
        img_u8 = img.contiguous()
        x = H.u8_to_nhwc(img_u8, dtype)  # preprocess_batch's float() / 255 (detect/train.py:59) + layout, one kernel
    else:
        x = H.to_nhwc(img.float().contiguous(), dtype)
    ys = []
    for m in model.model:
        if m.f != -1:
            x = ys[m.f] if isinstance(m.f, int) else [x if j == -1 else ys[j] for j in m.f]
        x = module_train(m, x, first=(m.i == 0), img_u8=img_u8 if m.i == 0 else None)
        ys.append(x if m.i in model.save else None)
    return x
