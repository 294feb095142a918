"""Block-level modules of the Drone-YOLO path (reference: ultralytics/nn/modules/block.py).

``DFL`` (:58-76), ``SPPF`` (:172-191), ``C2f`` (:227-249), ``Bottleneck`` (:337-350) and the fork's
own ``conv_bn`` / ``SEBlock`` / ``RepVGGBlock`` (:1365-1490).  chunk / cat inside C2f and SPPF are
done by construction: every producer writes its channel slice of one NHWC buffer.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import hip_ops as H
from .conv import Conv, _PackedMixin, _train_forward, fold_conv_bn

__all__ = ("DFL", "SPPF", "C2f", "Bottleneck", "RepVGGBlock", "SEBlock", "conv_bn")


class DFL(nn.Module):
    """Integral of the distribution-focal-loss bins — reference block.py:58-76.

    Holds the frozen arange(c1) 1x1 conv for state-dict compatibility (``dfl.conv.weight``); the
    softmax-expectation itself is fused into ``dy_detect_decode``.
    """

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        raise RuntimeError("DFL is fused into Detect's decode kernel (dy_detect_decode); call Detect instead")


class Bottleneck(nn.Module):
    """x + cv2(cv1(x)) when shortcut and c1 == c2 — reference block.py:337-350."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        if self.training:
            return _train_forward(self, "bottleneck_train", x, out=out)
        # the residual add rides in cv2's epilogue (after its SiLU, as in the reference expression)
        return self.cv2(self.cv1(x), out=out, residual=x if self.add else None)


class C2f(nn.Module):
    """CSP bottleneck with 2 convolutions, 'faster' variant — reference block.py:227-249."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    fuse_block = True  # one-kernel execution where dy_c2f_fused is built for the shape (the stride-4 backbone and neck blocks)

    def _packed_block(self, dtype, device):
        convs = (self.cv1, self.m[0].cv1, self.m[0].cv2, self.cv2)
        key = (dtype, str(device), H.scaled_domain(),
               tuple((c.conv.weight.data_ptr(), c.conv.weight._version, c.bn.weight._version, c.bn.running_var._version) for c in convs))
        cache = self.__dict__.get("_block_cache")
        if cache is None or cache[0] != key:
            folded = [H.domain_fold(*fold_conv_bn(c.conv.weight, c.conv.bias, c.bn), True)[:2] for c in convs]
            cache = (key, H.PackedC2f(*folded, shortcut=self.m[0].add, dtype=dtype, device=device, act_l2e=H.scaled_domain()))
            self.__dict__["_block_cache"] = cache
        return cache[1]

    def forward(self, x, out=None, **kw):
        """cv1 -> [y0 | y1]; y_{i+2} = m_i(y_{i+1}); cv2(cat(y)).  One buffer holds every y_i.

        ``kw`` (x2= / up2x=) is forwarded to cv1 so that a Concat(+Upsample) in front of this block
        can be folded into cv1's gather.
        """
        if self.training:
            return _train_forward(self, "c2f_train", x, out=out, **kw)
        if kw.pop("fp8_internal", False):
            return self._forward_fp8_internal(x, out)
        if self.fuse_block and len(self.m) == 1:
            cout = self.cv2.conv.out_channels
            if not kw and H.c2f_fused_supported(x.shape[1], self.c, cout, 1, x.dtype):
                return H.c2f_fused(x, self._packed_block(x.dtype, x.device), out=out)
            x2 = kw.get("x2")
            if kw.get("up2x") and x2 is not None and len(kw) == 2 and H.c2f_fused_supported(x.shape[1] + x2.shape[1], self.c, cout, 1, x.dtype, cin_lo=x.shape[1]):
                return H.c2f_fused(x2, self._packed_block(x.dtype, x.device), out=out, x_lo=x)  # Upsample + Concat + C2f in one launch
        n, _, hb, wb = x.shape
        h, w = (2 * hb, 2 * wb) if kw.get("up2x") else (hb, wb)
        c = self.c
        ybuf = H.alloc_nhwc(n, (2 + len(self.m)) * c, h, w, x.dtype, x.device)
        self.cv1(x, out=ybuf[:, : 2 * c], **kw)
        for i, m in enumerate(self.m):
            m(ybuf[:, (1 + i) * c : (2 + i) * c], out=ybuf[:, (2 + i) * c : (3 + i) * c])
        return self.cv2(ybuf, out=out)


def _c2f_fp8_internal(self, x, out=None):
    """C2f with its internals in e4m3 (mixed plan of BASELINE config 5, nn/tasks.py::_predict_layers): cv1 reads the 16-bit input and
    writes fp8 ([y0 | y1] in ONE fp8 buffer), every Bottleneck convolution runs fp8 -> fp8 on the block-scaled MFMA, cv2 reads the
    fp8 buffer and writes the 16-bit output — the block's boundary types are its caller's (csrc/conv_gemm_fk.hip, y_dtype1)."""
    n, _, h, w = x.shape
    c = self.c
    ybuf = H.alloc_nhwc(n, (2 + len(self.m)) * c, h, w, H.FP8, x.device)
    self.cv1(x, out=ybuf[:, : 2 * c], out_dtype=H.FP8)
    for i, m in enumerate(self.m):
        m(ybuf[:, (1 + i) * c : (2 + i) * c], out=ybuf[:, (2 + i) * c : (3 + i) * c])
    return self.cv2(ybuf, out=out, out_dtype=x.dtype)


C2f._forward_fp8_internal = _c2f_fp8_internal


class SPPF(nn.Module):
    """Spatial pyramid pooling (fast): cv1, three chained k x k max pools, cv2 — reference block.py:172-191."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.k = k
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)  # kept for repr / state parity only

    def forward(self, x, out=None):
        if self.training:
            return _train_forward(self, "sppf_train", x, out=out)
        n, _, h, w = x.shape
        c_ = self.cv1.conv.out_channels
        ybuf = H.alloc_nhwc(n, 4 * c_, h, w, x.dtype, x.device)
        self.cv1(x, out=ybuf[:, :c_])
        H.sppf_maxpool3(ybuf[:, :c_], ybuf[:, c_ : 2 * c_], ybuf[:, 2 * c_ : 3 * c_], ybuf[:, 3 * c_ :], self.k)
        return self.cv2(ybuf, out=out)


def conv_bn(in_channels, out_channels, kernel_size, stride, padding, groups=1):
    """Sequential(conv(bias=False), bn) with the reference's child names — block.py:1365-1372."""
    result = nn.Sequential()
    result.add_module("conv", nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, groups=groups, bias=False))
    result.add_module("bn", nn.BatchNorm2d(out_channels))
    return result


class SEBlock(nn.Module):
    """Squeeze-excite gate of RepVGGBlock(use_se=True) — reference block.py:1374-1391.

    Parameter container only: no Drone-YOLO YAML enables it (use_se defaults to False), so no
    kernel is built for it and ``forward`` refuses rather than falling back to eager PyTorch.
    """

    def __init__(self, input_channels, internal_neurons):
        super().__init__()
        self.down = nn.Conv2d(input_channels, internal_neurons, kernel_size=1, stride=1, bias=True)
        self.up = nn.Conv2d(internal_neurons, input_channels, kernel_size=1, stride=1, bias=True)
        self.input_channels = input_channels

    def forward(self, inputs):
        raise NotImplementedError("SEBlock (RepVGGBlock use_se=True) has no HIP kernel: not on the Drone-YOLO path")


class RepVGGBlock(_PackedMixin, nn.Module):
    """RepVGG block: SiLU(BN(conv3x3) + BN(conv1x1) + BN(identity)) — reference block.py:1393-1490.

    The three branches are folded into ONE 3x3 kernel and bias when the weights are packed
    (``get_equivalent_kernel_bias``, :1446-1478), so the device always runs the deploy form; the
    state dict keeps the training-time keys (``rbr_dense.conv.weight``, ``rbr_1x1.bn.*`` ...).
    """

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 padding_mode="zeros", deploy=False, use_se=False):
        super().__init__()
        if dilation != 1 or padding_mode != "zeros" or kernel_size != 3:
            raise NotImplementedError("RepVGGBlock is built for 3x3, dilation 1, zero padding")
        self.deploy = deploy
        self.groups = groups
        self.in_channels = in_channels
        self.stride, self.padding = stride, padding
        padding_11 = padding - kernel_size // 2
        self.nonlinearity = nn.SiLU()
        self.se = SEBlock(out_channels, internal_neurons=out_channels // 16) if use_se else nn.Identity()
        if deploy:
            self.rbr_reparam = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation=dilation,
                                         groups=groups, bias=True, padding_mode=padding_mode)
        else:
            self.rbr_identity = nn.BatchNorm2d(in_channels) if out_channels == in_channels and stride == 1 else None
            self.rbr_dense = conv_bn(in_channels, out_channels, kernel_size, stride, padding, groups=groups)
            self.rbr_1x1 = conv_bn(in_channels, out_channels, 1, stride, padding_11, groups=groups)

    # -- folding (reference block.py:1446-1478) ---------------------------------------------------
    def _fuse_bn_tensor(self, branch):
        if branch is None:
            return 0, 0
        if isinstance(branch, nn.Sequential):
            return fold_conv_bn(branch.conv.weight, None, branch.bn)
        input_dim = self.in_channels // self.groups  # identity branch: a 3x3 kernel with a centre 1
        k = torch.zeros((self.in_channels, input_dim, 3, 3), dtype=torch.float32, device=branch.weight.device)
        k[torch.arange(self.in_channels), torch.arange(self.in_channels) % input_dim, 1, 1] = 1.0
        return fold_conv_bn(k, None, branch)

    def get_equivalent_kernel_bias(self):
        k3, b3 = self._fuse_bn_tensor(self.rbr_dense)
        k1, b1 = self._fuse_bn_tensor(self.rbr_1x1)
        kid, bid = self._fuse_bn_tensor(getattr(self, "rbr_identity", None))
        k1 = torch.nn.functional.pad(k1, [1, 1, 1, 1]) if isinstance(k1, torch.Tensor) else 0
        return k3 + k1 + kid, b3 + b1 + bid

    def switch_to_deploy(self):
        """Replace the branches by the folded conv, as the reference's switch_to_deploy (:1421-1444)."""
        if hasattr(self, "rbr_1x1"):
            kernel, bias = self.get_equivalent_kernel_bias()
            d = self.rbr_dense.conv
            self.rbr_reparam = nn.Conv2d(d.in_channels, d.out_channels, d.kernel_size, d.stride, d.padding,
                                         dilation=d.dilation, groups=d.groups, bias=True)
            self.rbr_reparam.weight.data = kernel.to(d.weight.device)
            self.rbr_reparam.bias.data = bias.to(d.weight.device)
            for p in self.parameters():
                p.detach_()
            del self.rbr_dense, self.rbr_1x1
            if hasattr(self, "rbr_identity"):
                del self.rbr_identity
            self.deploy = True
            self.invalidate_packed()

    def _pack(self, dtype, device, cin_pad=None) -> H.PackedConv:
        if hasattr(self, "rbr_reparam"):
            w, b = self.rbr_reparam.weight, self.rbr_reparam.bias
        else:
            w, b = self.get_equivalent_kernel_bias()
        w, b, act = H.domain_fold(w, b, True, raw_input=getattr(self, "_raw_input", False))
        return H.PackedConv(w, b, self.stride, self.padding, self.groups, act, dtype, device, cin_pad=cin_pad)

    def forward(self, inputs, out=None):
        if not isinstance(self.se, nn.Identity):
            return self.se(inputs)  # raises: no SE kernel
        if self.training:
            return _train_forward(self, "repvgg_train", inputs, out=out)
        return H.conv2d(inputs, self._packed_for(inputs), out=out)
