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
        key = (dtype, str(device), H.scaled_domain(), tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in tails))
        cache = getattr(self, "_tail_cache", None)
        if cache is None or cache[0] != key:
            # (in the scaled activation domain the tails divide by log2 e: the logits the decode reads are in true units)
            pb = [H.pack_frag1x1(*H.domain_fold(s[-1].weight, s[-1].bias, False, raw_output=True)[:2], dtype, device) for s in self.cv2]
            pc = [H.pack_frag1x1(*H.domain_fold(s[-1].weight, s[-1].bias, False, raw_output=True)[:2], dtype, device) for s in self.cv3]
            cache = (key, pb, pc)
            self._tail_cache = cache
        return cache[1], cache[2]

    fuse_first = True  # run cv2[i][0] and cv3[i][0] (same input, both 3x3) as ONE convolution on the deep levels

    def _packed_first(self, i, dtype, device):
        """cv2[i][0] and cv3[i][0] stacked along cout, or None when the level is not one the stacking pays for.

        Both read the same feature map; stacked they make a cin -> c2 + c3 3x3 layer that reads it once: from 128 input
        channels up it runs in the deep-layer GEMM kernel (conv3x3_vgemm.hip), at 64 (the P2 level, the largest map of the
        model) in the register-weight kernel (conv3x3_hreg.hip, cout 128 = two 64-cout groups that share the halo through L2).
        Same operands and fp32 accumulation per output; only the summation order over K differs from the two-launch path.
        """
        a, b = self.cv2[i][0], self.cv3[i][0]
        if not (self.fuse_first and isinstance(a, Conv) and isinstance(b, Conv) and not isinstance(a, DWConv) and not isinstance(b, DWConv)):
            return None
        ca, cb = a.conv, b.conv
        if not (ca.kernel_size == cb.kernel_size == (3, 3) and ca.stride == cb.stride == (1, 1) and ca.groups == cb.groups == 1
                and ca.in_channels == cb.in_channels and ca.in_channels >= 64 and (ca.out_channels + cb.out_channels) % 128 == 0
                and ca.out_channels % 8 == 0 and isinstance(a.act, nn.SiLU) and isinstance(b.act, nn.SiLU)):
            return None
        srcs = [ca.weight, a.bn.weight, a.bn.bias, a.bn.running_mean, a.bn.running_var, cb.weight, b.bn.weight, b.bn.bias, b.bn.running_mean, b.bn.running_var]
        key = (dtype, str(device), H.fp8_act_scale() if dtype == H.FP8 else None, H.scaled_domain(), tuple((t.data_ptr(), t._version) for t in srcs))
        cache = self.__dict__.setdefault("_first_cache", {})
        hit = cache.get(i)
        if hit is None or hit[0] != key:
            from .conv import fold_conv_bn

            wa, ba = fold_conv_bn(ca.weight, ca.bias, a.bn)
            wb, bb = fold_conv_bn(cb.weight, cb.bias, b.bn)
            ws, bs, act = H.domain_fold(torch.cat((wa, wb), 0), torch.cat((ba, bb), 0), True)
            hit = (key, H.PackedConv(ws, bs, 1, 1, 1, act, dtype, device), ca.out_channels)
            cache[i] = hit
        return hit[1], hit[2]

    def _trunks(self, i, x):
        """Outputs of the two branch trunks (all but the last 1x1 conv) of level i."""
        pf = self._packed_first(i, x.dtype, x.device)
        if pf is None:
            return self._run_trunk(self.cv2[i], x), self._run_trunk(self.cv3[i], x)
        pc, c2 = pf
        both = H.conv2d(x, pc)
        tb, tc = both[:, :c2], both[:, c2:]
        for m in list(self.cv2[i])[1:-1]:
            tb = m(tb)
        for m in list(self.cv3[i])[1:-1]:
            tc = m(tc)
        return tb, tc

    fuse_branch = True  # second 3x3 conv + 1x1 + decode of every branch in one kernel where dy_detect_branch_fused is built

    def _branches_fusable(self, dtype) -> bool:
        """Every level's branches end Conv(c, 64, 3) -> Conv(64, 64, 3) -> Conv2d(64, 4*reg_max | nc <= 16, 1) in 16-bit storage."""
        if not self.fuse_branch or self.reg_max != 16:
            return False
        for i in range(self.nl):
            for seq, kind in ((self.cv2[i], 1), (self.cv3[i], 2)):
                mods = list(seq)
                if len(mods) != 3 or not (isinstance(mods[0], Conv) and isinstance(mods[1], Conv)) or isinstance(mods[1], DWConv):
                    return False
                c1 = mods[1].conv
                if not (c1.kernel_size == (3, 3) and c1.stride == (1, 1) and c1.groups == 1 and isinstance(mods[1].act, nn.SiLU)):
                    return False
                if not H.branch_fused_supported(c1.in_channels, c1.out_channels, mods[2].out_channels, kind, self.nc, self.reg_max, dtype):
                    return False
        return True

    def _forward_branch_fused(self, x):
        """Inference with each branch's second 3x3 conv, its 1x1 conv and its share of the decode (+ NMS candidate filter) in ONE
        kernel per (level, branch): the trunk outputs never reach HBM (dy_detect_branch_fused, csrc/conv3x3_hhead.hip)."""
        n = x[0].shape[0]
        A = sum(t.shape[2] * t.shape[3] for t in x)
        dtype, dev = x[0].dtype, x[0].device
        pred = torch.empty((n, 4 + self.nc, A), dtype=torch.float32, device=dev)
        fused = getattr(self, "fused_nms", None)
        bufs, conf, mask = None, 0.25, None
        if fused is not None:
            make_bufs, conf, mask = fused
            bufs = make_bufs(n, A)
            H.nms_reset_counts(bufs)
        pb, pc = self._packed_tail(dtype, dev)
        a0 = 0
        for i in range(self.nl):
            pf = self._packed_first(i, dtype, dev)
            if pf is None:
                tb, tc = self.cv2[i][0](x[i]), self.cv3[i][0](x[i])
            else:
                both = H.conv2d(x[i], pf[0])
                tb, tc = both[:, : pf[1]], both[:, pf[1] :]
            H.detect_branch_fused(tb, self.cv2[i][1]._packed_for(tb), pb[i][0], pb[i][1], 1, self.nc, self.reg_max, float(self.stride[i]), pred, a0)
            H.detect_branch_fused(tc, self.cv3[i][1]._packed_for(tc), pc[i][0], pc[i][1], 2, self.nc, self.reg_max, float(self.stride[i]), pred, a0,
                                  nms_bufs=bufs, conf_thres=conf, classes_mask=mask)
            a0 += x[i].shape[2] * x[i].shape[3]
        return pred

    def _forward_fused(self, x):
        """Inference with the branch tails, decode and NMS filter in one launch (dy_detect_head_decode)."""
        if self._branches_fusable(x[0].dtype):
            return self._forward_branch_fused(x)
        tr = [self._trunks(i, x[i]) for i in range(self.nl)]
        xb, xc = [t[0] for t in tr], [t[1] for t in tr]
        pb, pc = self._packed_tail(xb[0].dtype, xb[0].device)
        fused = getattr(self, "fused_nms", None)
        kw = {}
        if fused is not None:
            make_bufs, conf, mask = fused
            A = sum(f.shape[2] * f.shape[3] for f in xb)
            kw = dict(nms_bufs=make_bufs(xb[0].shape[0], A), conf_thres=conf, classes_mask=mask)
        return H.detect_head_decode(xb, xc, pb, pc, [float(s) for s in self.stride], self.nc, self.reg_max, **kw)

    tail_dtype = torch.float16  # storage type of the Detect branches behind an fp8 trunk (DY_FP8 inputs): see _forward_fp8_trunk

    def _forward_fp8_trunk(self, x):
        """Levels that arrive in fp8 (BASELINE config 5: an e4m3 trunk): the FIRST 3x3 convolution of each branch reads the fp8 map on the
        block-scaled MFMA and writes ``tail_dtype`` (float16) — from there the branch is the 16-bit path: second 3x3, then the 1x1 tails +
        DFL / sigmoid decode + candidate filter in one launch where that kernel is built.  The scores and box bins, where a rounding step
        decides a detection, never pass through a 3-bit mantissa.  Inference only."""
        if self.training or not self.legacy:
            raise NotImplementedError("Detect on fp8 inputs is built for inference with the v8 (legacy) class branch")
        td = self.tail_dtype
        xb, xc = [], []
        for i in range(self.nl):
            kw = {"out_dtype": td} if x[i].dtype == H.FP8 else {}
            tb, tc = self.cv2[i][0](x[i], **kw), self.cv3[i][0](x[i], **kw)
            for m in list(self.cv2[i])[1:-1]:
                tb = m(tb)
            for m in list(self.cv3[i])[1:-1]:
                tc = m(tc)
            xb.append(tb), xc.append(tc)
        fused = getattr(self, "fused_nms", None)
        kw = {}
        if fused is not None:
            make_bufs, conf, mask = fused
            kw = dict(nms_bufs=make_bufs(xb[0].shape[0], sum(f.shape[2] * f.shape[3] for f in xb)), conf_thres=conf, classes_mask=mask)
        if self.fuse_tail and H.head_decode_supported(self.cv2[0][-1].in_channels, self.cv3[0][-1].in_channels, self.nc, self.reg_max, td):
            pb, pc = self._packed_tail(td, xb[0].device)
            y = H.detect_head_decode(xb, xc, pb, pc, [float(s) for s in self.stride], self.nc, self.reg_max, **kw)
            return y if self.export else (y, None)
        nb, ld = self.reg_max * 4, (self.no + 3) // 4 * 4
        feats = []
        for i in range(self.nl):
            n, _, h, w = xb[i].shape
            buf = H.alloc_nhwc(n, self.no, h, w, torch.float32, xb[i].device, ld=ld)
            self.cv2[i][-1](xb[i], out=buf[:, :nb], out_f32=True)
            self.cv3[i][-1](xc[i], out=buf[:, nb:], out_f32=True)
            feats.append(buf)
        y = H.detect_decode(feats, [float(s) for s in self.stride], self.nc, self.reg_max, **kw)
        return y if self.export else (y, feats)

    def forward(self, x):
        if any(t.dtype == H.FP8 for t in x) and self.tail_dtype is not None:
            return self._forward_fp8_trunk(x)
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
