"""Convolution-level modules of the Drone-YOLO path on libdyolo kernels.

Same class names, constructor signatures and state-dict keys as the reference
(ultralytics/nn/modules/conv.py): ``Conv`` (:37-55), ``DWConv`` (:102-107), ``Concat`` (:323-333),
``autopad`` (:28-34).  Parameters live in ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` children so
checkpoints keep their key names (``conv.weight``, ``bn.running_mean`` ...), but those children
are never *called*: ``forward`` folds BatchNorm into the weights (utils/torch_utils.py:242-269),
packs them once per dtype and launches ``dy_conv2d_nhwc``.

Every forward takes NHWC-view tensors (see hip_ops) and an optional ``out=`` view so that the
graph executor can make producers write straight into Concat buffers.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from ... import hip_ops as H

__all__ = ("autopad", "Conv", "DWConv", "Concat", "Upsample", "fold_conv_bn")


def autopad(k, p=None, d=1):
    """'same' padding for kernel k and dilation d (reference conv.py:28-34)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def fold_conv_bn(conv_weight: torch.Tensor, conv_bias: Optional[torch.Tensor], bn: nn.BatchNorm2d):
    """W' = diag(g/sqrt(var+eps)) W,  b' = beta + (b - mean) g/sqrt(var+eps)  (torch_utils.py:242-269)."""
    w = conv_weight.detach().float()
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    b0 = torch.zeros(w.shape[0], device=w.device) if conv_bias is None else conv_bias.detach().float()
    return w * scale.view(-1, 1, 1, 1), bn.bias.detach().float() + (b0 - bn.running_mean.detach().float()) * scale


class _PackedMixin:
    """Lazy per-dtype cache of packed weights; dropped whenever parameters may have changed."""

    def _pack_cache(self) -> dict:
        """The per-dtype pack cache, emptied when any parameter / buffer of this module was re-homed or edited in place
        (storage identity + torch's version counter) since the packs were made."""
        sig = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        cache = self.__dict__.get("_packed")
        if cache is None or cache.get("__sig__") != sig:
            cache = self.__dict__["_packed"] = {"__sig__": sig}
        return cache

    def _packed_for(self, x: torch.Tensor) -> H.PackedConv:
        cache = self._pack_cache()
        # an fp8 pack bakes the activation scale in; every pack the activation domain it was folded for (H.scaled_activations)
        key = (x.dtype, x.device, x.shape[1], H.fp8_act_scale() if x.dtype == H.FP8 else None, H.scaled_domain())
        pc = cache.get(key)
        if pc is None:
            pc = cache[key] = self._pack(x.dtype, x.device)
            if pc.groups == 1 and 0 < x.shape[1] - pc.cin < H.chan_gran(x.dtype):
                # zero-padded input channels (image input padded to one 16-byte chunk): pad the taps to match
                pc = cache[key] = self._pack(x.dtype, x.device, cin_pad=x.shape[1])
        return pc

    def invalidate_packed(self) -> None:
        self.__dict__.pop("_packed", None)

    def train(self, mode: bool = True):
        if mode != self.training:  # a training phase writes the parameters through raw pointers: packs made before it are stale
            self.invalidate_packed()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        before = [(t.data_ptr(), t.dtype, t.device) for t in list(self.parameters()) + list(self.buffers())]
        out = super()._apply(fn, *args, **kwargs)
        if before != [(t.data_ptr(), t.dtype, t.device) for t in list(self.parameters()) + list(self.buffers())]:
            self.invalidate_packed()  # .to(device) / .half(): a no-op move (second predictor on the same model) keeps the packs
        return out


def _train_forward(m: nn.Module, fn_name: str, x, **kw):
    """``module.train()(x)`` on a module of its own (reference conv.py:49-51, block.py:1480-1490, :237-242: batch-statistics BatchNorm,
    running statistics updated, autograd graph): the same autograd ops the model-level executor uses (nn/train_forward.py).  ``x`` is an
    NHWC view on the device, as for eval; the eval-only launch options (out= slices, fused residual / Concat gathers) do not exist here."""
    extra = {k: v for k, v in kw.items() if v is not None and v is not False}
    if extra:
        raise NotImplementedError(f"{type(m).__name__}: {sorted(extra)} are launch options of the eval path; a training-mode forward takes the input only")
    from .. import train_forward as TF

    return getattr(TF, fn_name)(m, x)


class Conv(_PackedMixin, nn.Module):
    """Conv2d(bias=False) + BatchNorm2d + SiLU, args (c1, c2, k, s, p, g, d, act) — reference conv.py:37-55."""

    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if d != 1:
            raise NotImplementedError("dilated convolutions are not on the Drone-YOLO path")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        if not isinstance(self.act, (nn.SiLU, nn.Identity)):
            raise NotImplementedError("only SiLU / identity activations are built into the conv epilogue")

    def _pack(self, dtype, device, cin_pad=None) -> H.PackedConv:
        w, b = fold_conv_bn(self.conv.weight, self.conv.bias, self.bn) if hasattr(self, "bn") else (
            self.conv.weight, self.conv.bias)
        c = self.conv
        if b is None:
            b = torch.zeros(w.shape[0], device=w.device)
        # _raw_input: set by the model executor on the layer that reads the image (its input is not in the scaled domain)
        w, b, act = H.domain_fold(w, b, isinstance(self.act, nn.SiLU), raw_input=getattr(self, "_raw_input", False))
        return H.PackedConv(w, b, c.stride[0], c.padding[0], c.groups, act, dtype, device, cin_pad=cin_pad)

    def forward(self, x, out=None, residual=None, **kw):
        if self.training:
            return _train_forward(self, "conv_train", x, out=out, residual=residual, **kw)
        return H.conv2d(x, self._packed_for(x), out=out, residual=residual, **kw)

    forward_fuse = forward  # BN is always folded on this path

    def is_stem(self) -> bool:
        """First-layer shape the fused image kernel covers: Conv(c1<=3, c2<=80, 3, 2) with padding 1."""
        c = self.conv
        return (c.kernel_size == (3, 3) and c.stride == (2, 2) and c.padding == (1, 1) and c.groups == 1
                and c.in_channels * 9 <= 32 and c.out_channels <= 80)

    def forward_stem(self, im, dtype, out=None, mark_input=False):
        """fp32 NCHW image -> this layer's NHWC output in ``dtype`` (layout cast + conv + BN + SiLU in one kernel)."""
        if self.training:
            raise NotImplementedError("Conv.forward_stem is the eval path's fused image kernel; training reads the image through model.forward_train")
        cache = self._pack_cache()
        key = ("stem", dtype, im.device, H.scaled_domain())
        ps = cache.get(key)
        if ps is None:
            w, b = fold_conv_bn(self.conv.weight, self.conv.bias, self.bn)
            w, b, act = H.domain_fold(w, b, isinstance(self.act, nn.SiLU), raw_input=True)  # the image is never in the scaled domain
            ps = cache[key] = H.PackedStem(w, b, act, dtype, im.device)
        return H.stem_conv(im, ps, out=out, mark_input=mark_input)


class DWConv(Conv):
    """Depth-wise convolution, g = gcd(c1, c2) — reference conv.py:102-107."""

    def __init__(self, c1, c2, k=1, s=1, d=1, act=True):
        super().__init__(c1, c2, k, s, g=math.gcd(c1, c2), d=d, act=act)


class PlainConv2d(_PackedMixin, nn.Conv2d):
    """nn.Conv2d with bias and no activation (the last layer of each Detect branch, head.py:43-57)."""

    def _pack(self, dtype, device, cin_pad=None) -> H.PackedConv:
        # the last layer of a Detect branch leaves the scaled activation domain: its logits are in true units (weights / log2 e)
        w, b, act = H.domain_fold(self.weight, self.bias, False, raw_output=True)
        return H.PackedConv(w, b, self.stride[0], self.padding[0], self.groups, act, dtype, device, cin_pad=cin_pad, for_out_f32=True)

    def forward(self, x, out=None, out_f32=False):
        return H.conv2d(x, self._packed_for(x), out=out, out_f32=out_f32)


class Concat(nn.Module):
    """Channel concatenation — reference conv.py:323-333.

    When every input already is the right channel slice of one buffer (the graph executor arranges
    that), the buffer is returned as is; otherwise the inputs are copied with ``dy_copy_nhwc``.
    """

    def __init__(self, dimension=1):
        super().__init__()
        if dimension != 1:
            raise NotImplementedError("Concat is built for the channel dimension only")
        self.d = dimension

    def forward(self, x, out=None):
        n, _, h, w = x[0].shape
        ctot = sum(t.shape[1] for t in x)
        if out is None:
            out = H.alloc_nhwc(n, ctot, h, w, x[0].dtype, x[0].device)
        op, ld = H.view_params(out)
        es = out.element_size()
        c0 = 0
        for t in x:
            c = t.shape[1]
            tp, tld = H.view_params(t)
            if not (tp == op + c0 * es and tld == ld):
                H.copy_nhwc(t, out[:, c0 : c0 + c])
            c0 += c
        return out


class Upsample(nn.Module):
    """nn.Upsample(None, 2, 'nearest') of the model YAMLs (yolov8-p2-repvgg.yaml:30,34,38)."""

    def __init__(self, size=None, scale_factor=None, mode="nearest"):
        super().__init__()
        if size is not None or scale_factor not in (2, 2.0) or mode != "nearest":
            raise NotImplementedError("only nn.Upsample(None, 2, 'nearest') is on the Drone-YOLO path")
        self.size, self.scale_factor, self.mode = size, scale_factor, mode

    def forward(self, x, out=None):
        return H.upsample2x(x, out=out)
