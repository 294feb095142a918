"""Detect head of the Drone-YOLO path (reference: ultralytics/nn/modules/head.py:21-172)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import hip_ops as H
from .block import DFL
from .conv import Conv, DWConv, PlainConv2d

__all__ = ("Detect",)


class Detect(nn.Module):
    """YOLO Detect head: per level box branch cv2 and class branch cv3, then decode.

    Same constructor, attributes and state-dict keys as the reference (head.py:34-61).  Forward
    (head.py:64-74): per level the two branches write their logits into ONE fp32 NHWC buffer
    (box bins in channels [0, 4*reg_max), classes after them), which *is* ``cat(cv2(x), cv3(x), 1)``
    of the reference seen through an NHWC view.  Eval returns ``(y, x)`` with ``y`` the decoded
    (N, 4+nc, A) tensor from ``dy_detect_decode``; training mode returns the raw list ``x``.
    """

    dynamic = False
    export = False
    format = None
    end2end = False
    max_det = 300
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)
    legacy = False

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), PlainConv2d(c2, 4 * self.reg_max, 1)) for x in ch
        )
        self.cv3 = (
            nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), PlainConv2d(c3, self.nc, 1)) for x in ch)
            if self.legacy
            else nn.ModuleList(
                nn.Sequential(
                    nn.Sequential(DWConv(x, x, 3), Conv(x, c3, 1)),
                    nn.Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)),
                    PlainConv2d(c3, self.nc, 1),
                )
                for x in ch
            )
        )
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()

    @staticmethod
    def _run(seq, x, out):
        """Run one branch; its last (plain 1x1) conv writes fp32 logits into ``out``."""
        mods = list(seq)
        for m in mods[:-1]:
            if isinstance(m, nn.Sequential):
                for mm in m:
                    x = mm(x)
            else:
                x = m(x)
        return mods[-1](x, out=out, out_f32=True)

    fuse_tail = False  # opt-in (the predictor sets it): eval returns (y, None), the raw maps are never materialised

    @staticmethod
    def _run_trunk(seq, x):
        """All but the last (plain 1x1) conv of one branch."""
        for m in list(seq)[:-1]:
            if isinstance(m, nn.Sequential):
                for mm in m:
                    x = mm(x)
            else:
                x = m(x)
        return x

    def _packed_tail(self, dtype, device):
        tails = [s[-1] for s in self.cv2] + [s[-1] for s in self.cv3]
        key = (dtype, str(device), tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in tails))
        cache = getattr(self, "_tail_cache", None)
        if cache is None or cache[0] != key:
            pb = [H.pack_frag1x1(s[-1].weight, s[-1].bias, dtype, device) for s in self.cv2]
            pc = [H.pack_frag1x1(s[-1].weight, s[-1].bias, dtype, device) for s in self.cv3]
            cache = (key, pb, pc)
            self._tail_cache = cache
        return cache[1], cache[2]

    def _forward_fused(self, x):
        """Inference with the branch tails, decode and NMS filter in one launch (dy_detect_head_decode)."""
        xb = [self._run_trunk(self.cv2[i], x[i]) for i in range(self.nl)]
        xc = [self._run_trunk(self.cv3[i], x[i]) for i in range(self.nl)]
        pb, pc = self._packed_tail(xb[0].dtype, xb[0].device)
        fused = getattr(self, "fused_nms", None)
        kw = {}
        if fused is not None:
            make_bufs, conf, mask = fused
            A = sum(f.shape[2] * f.shape[3] for f in xb)
            kw = dict(nms_bufs=make_bufs(xb[0].shape[0], A), conf_thres=conf, classes_mask=mask)
        return H.detect_head_decode(xb, xc, pb, pc, [float(s) for s in self.stride], self.nc, self.reg_max, **kw)

    def forward(self, x):
        if self.fuse_tail and not self.training and H.head_decode_supported(
                self.cv2[0][-1].in_channels, self.cv3[0][-1].in_channels, self.nc, self.reg_max, x[0].dtype):
            y = self._forward_fused(x)
            return y if self.export else (y, None)
        nb = self.reg_max * 4
        ld = (self.no + 3) // 4 * 4  # keep every pixel row 16-byte aligned for the fp32 vector paths
        feats = []
        for i in range(self.nl):
            n, _, h, w = x[i].shape
            buf = H.alloc_nhwc(n, self.no, h, w, torch.float32, x[i].device, ld=ld)
            self._run(self.cv2[i], x[i], buf[:, :nb])
            self._run(self.cv3[i], x[i], buf[:, nb:])
            feats.append(buf)
        if self.training:
            return feats
        fused = getattr(self, "fused_nms", None)  # set by the predictor: (NmsBuffers factory, conf, classes mask)
        kw = {}
        if fused is not None:
            make_bufs, conf, mask = fused
            A = sum(f.shape[2] * f.shape[3] for f in feats)
            kw = dict(nms_bufs=make_bufs(feats[0].shape[0], A), conf_thres=conf, classes_mask=mask)
        y = H.detect_decode(feats, [float(s) for s in self.stride], self.nc, self.reg_max, **kw)
        return y if self.export else (y, feats)

    def bias_init(self):
        """box bias 1.0; cls bias log(5 / nc / (640/s)^2) — reference head.py:133-144."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / float(s)) ** 2)
            a[-1].invalidate_packed()
            b[-1].invalidate_packed()
        self._tail_cache = None
