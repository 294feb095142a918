"""Training-mode building blocks: every compute step is a libdyolo kernel; ``torch.autograd.Function`` only records the
graph (which gradient feeds which op) the way the reference relies on autograd (engine/trainer.py:381-389).

Reference semantics: Conv.forward in training = SiLU(BatchNorm_batchstats(conv(x))) (nn/modules/conv.py:49-51),
RepVGGBlock.forward = SiLU(BN(conv3x3(x)) + BN(conv1x1(x))) (nn/modules/block.py:1480-1490), Detect.forward's training
return (nn/modules/head.py:64-72), C2f / SPPF / Bottleneck / Concat / Upsample (block.py:172-350, conv.py:323-333).

Activations are NHWC-view tensors in the storage dtype (bf16 / fp16 / fp32); parameters are the modules' fp32 master
tensors on the device; gradients of activations come back in the storage dtype, of parameters in fp32.
"""
from __future__ import annotations

from typing import List, Sequence

import os

import torch

from .. import hip_ops as H


def as_nhwc(t: torch.Tensor) -> torch.Tensor:
    """A gradient handed over by autograd as an NHWC view with whole 16-byte chunks per pixel (no copy when it already is)."""
    n, c, h, w = t.shape
    st = t.stride()
    es = t.element_size()
    if st[1] == 1 and st[3] % (16 // es) == 0 and st[2] == w * st[3] and st[0] == h * st[2] and t.data_ptr() % 16 == 0 and st[3] >= c:
        return t
    out = H.alloc_nhwc(n, c, h, w, t.dtype, t.device)
    out.copy_(t)
    return out


CONV_STATS = [os.environ.get("DYOLO_CONV_STATS", "1") != "0"]  # BatchNorm statistics from convolution epilogues (conv_bn_fwd)
STATS_LOG = [None]  # tools/train_stats_census.py: a list here collects (weight shape, stride, input shape, kernel, slots) of every conv_bn_fwd


def conv_bn_fwd(x, weight, gamma, beta, bn, stride, pad, act, out=None, stem_u8=None):
    """z = conv2d(x, weight), y = act(BN_batchstats(z)) (y into ``out`` when given): returns (z, BnState, y)."""
    dtype, dev = x.dtype, x.device
    # batch statistics from the convolution's epilogue where the launched kernel has one (dy_conv_desc.bn_stats): partial sums straight
    # into the BatchNorm's workspace, no reduction pass over z
    st = H.BnState(weight.shape[0], dev)
    if stem_u8 is not None:  # the image stem straight from the uint8 batch (x, its NHWC form, only serves the weight gradient)
        z, slabs = H.stem_conv_u8(stem_u8, weight, dtype), 0
    else:
        cin_pad = x.shape[1] if x.shape[1] != weight.shape[1] else None  # image padded to one chunk
        pc = H.PackedConv(weight, H.zero_bias(weight.shape[0], dev), stride, pad, 1, False, dtype, dev, cin_pad=cin_pad)
        z = H.conv2d(x, pc, bn_stats=st if CONV_STATS[0] else None)
        slabs = H.conv_stats_written() if CONV_STATS[0] else 0
        if STATS_LOG[0] is not None:
            STATS_LOG[0].append((tuple(weight.shape), stride, tuple(x.shape), H.last_kernel_name(), slabs))
    y = H.bn_train_fwd(z, gamma, beta, st, act, eps=bn.eps, momentum=bn.momentum, running_mean=bn.running_mean, running_var=bn.running_var, out=out,
                       partial_slabs=slabs)
    return z, st, y


# Seed of the backward pass, deferred: under ``lazy_head_seed`` (the trainer's backward) DetectionLossFn.backward hands its stored
# gradients on without the multiply by d loss_out / d total and leaves the factor here, keyed by the gradient's address; the consumer
# (HeadTail.backward) takes it out again.  The trainer checks that nothing is left over: an entry nobody consumed means a gradient
# went on unscaled.
LAZY_SEED: dict = {}
_LAZY = [False]


class lazy_head_seed:
    def __enter__(self):
        _LAZY[0] = os.environ.get("DYOLO_LAZY_SEED", "1") != "0"
        LAZY_SEED.clear()

    def __exit__(self, et, ev, tb):
        _LAZY[0] = False
        left = len(LAZY_SEED)
        LAZY_SEED.clear()
        if et is None and left:
            raise RuntimeError(f"lazy_head_seed: {left} head gradient(s) were passed on without their scale (no HeadTail consumed them)")
        return False


_SINK_ARMED = [False]  # set by the trainer around ITS forward + backward only: a backward run by anybody else keeps autograd's gradients
_SINK_NOTE = [None]  # callback(param): this parameter's sink slot was just handed out, i.e. the kernels that write it are launched next


class sink_armed:
    """``note``: called with every parameter whose sink slot a backward function takes (parallel.GradBuckets counts its buckets down
    with it: when the NEXT slot is taken, the previous parameter's gradient kernels are already in the stream)."""

    def __init__(self, note=None):
        self.note = note

    def __enter__(self):
        _SINK_ARMED[0] = True
        _SINK_NOTE[0] = self.note

    def __exit__(self, *exc):
        _SINK_ARMED[0] = False
        _SINK_NOTE[0] = None
        return False


def grad_sink(p):
    """The flat fp32 slot a trainer set aside for this parameter's gradient of the current batch (``FlatState.enable_sink``), or
    None.  With a slot the backward kernels write there and return no gradient to autograd, and one ``dy_grad_sink_flush`` per
    batch (per gradient bucket with several ranks) adds every slot to ``param.grad`` — instead of one AccumulateGrad add (and one
    zero-filled temporary) per parameter."""
    if not _SINK_ARMED[0]:
        return None
    s = getattr(p, "_dy_sink", None)
    if s is not None and _SINK_NOTE[0] is not None:
        _SINK_NOTE[0](p)
    return s


def _commits_sink(fn):
    """Decorator for a Function's ``backward``: when it returns, every kernel that writes the sink slots it took is in the stream —
    tell the listener (``note(None)``), which may then flush / exchange a gradient bucket whose last parameter this was."""

    def backward(ctx, *grads):
        out = fn(ctx, *grads)
        if _SINK_ARMED[0] and _SINK_NOTE[0] is not None:
            _SINK_NOTE[0](None)
        return out

    return staticmethod(backward)


# BatchNorm-backward sums from the epilogue of the input-gradient convolution behind (dy_conv_desc.bnb_z): on unless
# DYOLO_BN_BEHIND=0.  Used where the structure says the intermediate has ONE consumer: cv1 -> cv2 of a Bottleneck inside C2fTrain and of
# the Detect branches (ConvBnAct2).
BN_BEHIND = [os.environ.get("DYOLO_BN_BEHIND", "1") != "0"]
BEHIND_LOG = [None]  # a list: (channels, h, w, slots) per candidate layer of one backward (tests, tools/train_stats_census.py)


def bn_behind_of(z, gamma, beta, st, act):
    """The ``H.BnBehind`` of a layer whose output feeds exactly one convolution, or None when the fusion is off or its operands do not fit."""
    if not BN_BEHIND[0] or z.dtype not in (torch.bfloat16, torch.float16) or gamma.dtype != torch.float32 or not gamma.is_contiguous() or not beta.is_contiguous():
        return None
    return H.BnBehind(z, gamma, beta, st, act)


def conv_bn_bwd(dy, x, z, weight, gamma, beta, st, stride, pad, act, need_dx=True, dx_out=None, dx_accumulate=None, behind=None, partial_slabs=0):
    """(dx, dw, dgamma, dbeta) of conv_bn_fwd.  ``dx_accumulate``: a gradient already held for x (another consumer's
    contribution), added in the dgrad epilogue; ``dx_out`` may be that same view (in place: a lane reads what it then overwrites).
    ``behind``: the ``H.BnBehind`` of the layer that produced x when this convolution is its only consumer (its ``slots`` are set by
    the input-gradient launch); ``partial_slabs``: the slots such a launch left for THIS layer's BatchNorm backward (dy then is exactly
    what that launch stored).  Parameter gradients that went to the trainer's sink come back as None."""
    cout, cin, k, _ = weight.shape
    sg, sb, sw = grad_sink(gamma), grad_sink(beta), grad_sink(weight)
    dz, dgamma, dbeta = H.bn_train_bwd(as_nhwc(dy), z, gamma, beta, st, act, dgamma=sg, dbeta=sb, partial_slabs=partial_slabs)
    if sg is not None:
        dgamma = None
    if sb is not None:
        dbeta = None
    if sw is not None and x.shape[1] == cin:
        H.conv_wgrad_into(x, dz, k, stride, pad, sw.view(cout, k, k, cin))  # on the side stream: overlaps the input gradient below
        dw = None
    elif sw is not None:
        # the image stem: its input is padded to one chunk, so the kernel's (cout, k, k, cin_pad) result is sliced into the sink slot.
        # r04: this gradient used to go back through autograd — the one AccumulateGrad node of a training step, created on the eager
        # warm-up step's stream and met again under hipGraph capture on another (PyTorch's "AccumulateGrad node's stream does not
        # match" warning, and the precondition of round 3's hipStreamEndCapture crash).  Now no parameter gradient passes through autograd.
        dwp = H.conv_wgrad(x, dz, k, stride, pad)  # (cout, cin_pad, k, k) view of the kernel's (cout, k, k, cin_pad) buffer
        sw.view(cout, k, k, cin).add_(dwp.permute(0, 2, 3, 1)[..., :cin])
        dw = None
    else:
        dw = H.conv_wgrad(x, dz, k, stride, pad)[:, :cin]
    dx = None
    if need_dx:
        dx = H.conv_dgrad(dz, H.pack_dgrad(weight, stride, x.dtype, x.device), stride, out=dx_out, accumulate=dx_accumulate, bn_behind=behind)
        if behind is not None and BEHIND_LOG[0] is not None:
            BEHIND_LOG[0].append((cin, x.shape[2], x.shape[3], H.last_kernel_name(), behind.slots))
    return dx, dw, dgamma, dbeta


class ConvBnAct(torch.autograd.Function):
    """y = act(BN_train(conv2d(x, w))) — dy_conv2d_nhwc + dy_bn_train_fwd; backward: dy_bn_train_bwd + wgrad + dgrad."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, stride, pad, act, need_dx, stem_u8=None):
        z, st, y = conv_bn_fwd(x, weight, gamma, beta, bn, stride, pad, act, stem_u8=stem_u8)
        ctx.save_for_backward(x, z, weight, gamma, beta)
        ctx.st, ctx.stride, ctx.pad, ctx.act, ctx.need_dx = st, stride, pad, act, need_dx
        return y

    @_commits_sink
    def backward(ctx, dy):
        x, z, weight, gamma, beta = ctx.saved_tensors
        dx, dw, dgamma, dbeta = conv_bn_bwd(dy, x, z, weight, gamma, beta, ctx.st, ctx.stride, ctx.pad, ctx.act, need_dx=ctx.need_dx)
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None


class ConvBnAct2(torch.autograd.Function):
    """Two Conv modules in a row whose intermediate nobody else reads — cv2[i][0] -> cv2[i][1] and cv3[i][0] -> cv3[i][1] of Detect
    (head.py:43-57), cv1 -> cv2 of a Bottleneck outside C2fTrain (block.py:337-350) — as ONE node, so that the second layer's
    input-gradient launch can leave the first layer's BatchNorm-backward sums (``conv_bn_bwd(behind=)``)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, bn1, bn2, geo, need_dx):
        (s1, p1, a1), (s2, p2, a2) = geo
        z1, st1, t = conv_bn_fwd(x, w1, g1, b1, bn1, s1, p1, a1)
        z2, st2, y = conv_bn_fwd(t, w2, g2, b2, bn2, s2, p2, a2)
        ctx.save_for_backward(x, z1, t, z2, w1, g1, b1, w2, g2, b2)
        ctx.st, ctx.geo, ctx.need_dx = (st1, st2), geo, need_dx
        return y

    @_commits_sink
    def backward(ctx, dy):
        x, z1, t, z2, w1, g1, b1, w2, g2, b2 = ctx.saved_tensors
        (s1, p1, a1), (s2, p2, a2) = ctx.geo
        behind = bn_behind_of(z1, g1, b1, ctx.st[0], a1)
        dt, dw2, dg2, db2 = conv_bn_bwd(dy, t, z2, w2, g2, b2, ctx.st[1], s2, p2, a2, behind=behind)
        dx, dw1, dg1, db1 = conv_bn_bwd(dt, x, z1, w1, g1, b1, ctx.st[0], s1, p1, a1, need_dx=ctx.need_dx, partial_slabs=behind.slots if behind is not None else 0)
        return dx, dw1, dg1, db1, dw2, dg2, db2, None, None, None, None


class C2fTrain(torch.autograd.Function):
    """A whole C2f block in training mode (block.py:237-242 with Bottleneck :337-350): cv1 -> chunk -> n Bottlenecks -> cat -> cv2.

    One buffer holds [y0 | y1 | y2 ...]: cv1's BatchNorm writes channels [0, 2c), every Bottleneck its own c channels, so the
    chunk and the cat are views.  Backward mirrors it on the gradient of that buffer: a Bottleneck's input gradient is
    accumulated into its slice by the dgrad epilogue (in place), the shortcut's by one dy_add_nhwc; what reaches cv1 is the
    [0, 2c) view.  The op-by-op graph spent 1.9 ms per step (B = 64) in Concat / chunk copies and 0.5 ms in autograd's sums of
    twice-used gradients.  ``params``: (weight, bn.weight, bn.bias) of cv1, then m[i].cv1, m[i].cv2 for every Bottleneck, then cv2.
    """

    @staticmethod
    def forward(ctx, x, block, *params):
        c, nb = block.c, len(block.m)
        convs = [block.cv1] + [cv for mm in block.m for cv in (mm.cv1, mm.cv2)] + [block.cv2]
        P = [params[3 * i : 3 * i + 3] for i in range(len(convs))]
        geo = [(cv.conv.stride[0], cv.conv.padding[0]) for cv in convs]
        n, _, h, w = x.shape
        buf = H.alloc_nhwc(n, (2 + nb) * c, h, w, x.dtype, x.device)
        saved, states = [], []
        z1, s1, _ = conv_bn_fwd(x, *P[0], convs[0].bn, *geo[0], True, out=buf[:, : 2 * c])
        saved.append(z1), states.append(s1)
        for i, mm in enumerate(block.m):
            yin, dst = buf[:, (1 + i) * c : (2 + i) * c], buf[:, (2 + i) * c : (3 + i) * c]
            za, sa, t = conv_bn_fwd(yin, *P[1 + 2 * i], convs[1 + 2 * i].bn, *geo[1 + 2 * i], True)
            if mm.add:
                zb, sb, u = conv_bn_fwd(t, *P[2 + 2 * i], convs[2 + 2 * i].bn, *geo[2 + 2 * i], True)
                H.add_nhwc(yin, u, out=dst)  # x + cv2(cv1(x))
            else:
                zb, sb, _ = conv_bn_fwd(t, *P[2 + 2 * i], convs[2 + 2 * i].bn, *geo[2 + 2 * i], True, out=dst)
            saved += [za, t, zb]
            states += [sa, sb]
        z2, s2, y = conv_bn_fwd(buf, *P[-1], convs[-1].bn, *geo[-1], True)
        saved.append(z2), states.append(s2)
        ctx.save_for_backward(x, buf, *saved, *params)
        ctx.states, ctx.geo, ctx.c, ctx.nb, ctx.add = states, geo, c, nb, [bool(mm.add) for mm in block.m]
        return y

    @_commits_sink
    def backward(ctx, dy):
        c, nb, geo, S = ctx.c, ctx.nb, ctx.geo, ctx.states
        t_all = ctx.saved_tensors
        x, buf, saved, params = t_all[0], t_all[1], t_all[2 : 4 + 3 * nb], t_all[4 + 3 * nb :]
        P = [params[3 * i : 3 * i + 3] for i in range(2 + 2 * nb)]
        grads = [None] * len(params)

        def put(i, dw, dg, db):
            grads[3 * i], grads[3 * i + 1], grads[3 * i + 2] = dw, dg, db

        dbuf, dw, dg, db = conv_bn_bwd(dy, buf, saved[-1], *P[-1], S[-1], *geo[-1], True)
        put(1 + 2 * nb, dw, dg, db)
        for i in reversed(range(nb)):
            za, t, zb = saved[1 + 3 * i : 4 + 3 * i]
            g, gin = dbuf[:, (2 + i) * c : (3 + i) * c], dbuf[:, (1 + i) * c : (2 + i) * c]
            if ctx.add[i]:
                H.add_nhwc(gin, g, out=gin)  # the shortcut hands the gradient straight to the Bottleneck's input
            # t = cv1's output feeds cv2 alone: the sums of cv1's BatchNorm backward come out of cv2's input-gradient launch where built
            behind = bn_behind_of(za, P[1 + 2 * i][1], P[1 + 2 * i][2], S[1 + 2 * i], True)
            dt, dw, dg, db = conv_bn_bwd(g, t, zb, *P[2 + 2 * i], S[2 + 2 * i], *geo[2 + 2 * i], True, behind=behind)
            put(2 + 2 * i, dw, dg, db)
            _, dw, dg, db = conv_bn_bwd(dt, buf[:, (1 + i) * c : (2 + i) * c], za, *P[1 + 2 * i], S[1 + 2 * i], *geo[1 + 2 * i], True, dx_out=gin, dx_accumulate=gin,
                                        partial_slabs=behind.slots if behind is not None else 0)
            put(1 + 2 * i, dw, dg, db)
        dx, dw, dg, db = conv_bn_bwd(dbuf[:, : 2 * c], x, saved[0], *P[0], S[0], *geo[0], True, need_dx=ctx.needs_input_grad[0])
        put(0, dw, dg, db)
        return (dx, None, *grads)


class GroupedConvBnAct(torch.autograd.Function):
    """ConvBnAct for a grouped convolution (DWConv of the -sf YAML, conv.py:102-107): forward through the grouped kernel of
    dy_conv2d_nhwc, both gradients through dy_conv2d_grouped_bwd_nhwc."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, stride, pad, groups, act):
        dtype, dev = x.dtype, x.device
        pc = H.PackedConv(weight, H.zero_bias(weight.shape[0], dev), stride, pad, groups, False, dtype, dev)
        z = H.conv2d(x, pc)
        st = H.BnState(weight.shape[0], dev)
        y = H.bn_train_fwd(z, gamma, beta, st, act, eps=bn.eps, momentum=bn.momentum, running_mean=bn.running_mean, running_var=bn.running_var)
        ctx.save_for_backward(x, z, weight, gamma, beta)
        ctx.st, ctx.stride, ctx.pad, ctx.groups, ctx.act = st, stride, pad, groups, act
        return y

    @_commits_sink
    def backward(ctx, dy):
        x, z, weight, gamma, beta = ctx.saved_tensors
        sg, sb, sw = grad_sink(gamma), grad_sink(beta), grad_sink(weight)
        dz, dgamma, dbeta = H.bn_train_bwd(as_nhwc(dy), z, gamma, beta, ctx.st, ctx.act, dgamma=sg, dbeta=sb)
        dw, dx = H.conv_grouped_bwd(x, dz, weight.detach().contiguous(), ctx.stride, ctx.pad, ctx.groups)
        if sw is not None:  # r05: into the trainer's sink like every other parameter gradient (slot order (cout, k, k, cin / groups), as the dense layers')
            cout, cg, k, _ = weight.shape
            sw.view(cout, k, k, cg).add_(dw.permute(0, 2, 3, 1))
            dw = None
        return dx, dw, None if sg is not None else dgamma, None if sb is not None else dbeta, None, None, None, None, None


class RepVGGTrain(torch.autograd.Function):
    """y = SiLU(BN3(conv3x3 s(x)) + BN1(conv1x1 s(x))) (no identity branch: the model's RepVGG blocks are stride 2)."""

    @staticmethod
    def forward(ctx, x, w3, g3, b3, w1, g1, b1, bn3, bn1, stride):
        dtype, dev = x.dtype, x.device
        zero = lambda w: H.zero_bias(w.shape[0], dev)  # noqa: E731
        s3, s1 = H.BnState(w3.shape[0], dev), H.BnState(w1.shape[0], dev)
        st = CONV_STATS[0]  # batch statistics from the convolutions' epilogues where the launched kernel has one (conv_bn_fwd)
        z3 = H.conv2d(x, H.PackedConv(w3, zero(w3), stride, 1, 1, False, dtype, dev), bn_stats=s3 if st else None)
        n3 = H.conv_stats_written() if st else 0
        if STATS_LOG[0] is not None:
            STATS_LOG[0].append((tuple(w3.shape), stride, tuple(x.shape), H.last_kernel_name(), n3))
        z1 = H.conv2d(x, H.PackedConv(w1, zero(w1), stride, 0, 1, False, dtype, dev, halo=False), bn_stats=s1 if st else None)
        n1 = H.conv_stats_written() if st else 0
        if STATS_LOG[0] is not None:
            STATS_LOG[0].append((tuple(w1.shape), stride, tuple(x.shape), H.last_kernel_name(), n1))
        u3 = H.bn_train_fwd(z3, g3, b3, s3, False, eps=bn3.eps, momentum=bn3.momentum, running_mean=bn3.running_mean, running_var=bn3.running_var,
                            partial_slabs=n3)
        u = H.bn_train_fwd(z1, g1, b1, s1, False, eps=bn1.eps, momentum=bn1.momentum, running_mean=bn1.running_mean, running_var=bn1.running_var,
                           addend=u3, partial_slabs=n1)
        y = H.silu_fwd(u)
        ctx.save_for_backward(x, z3, z1, u, w3, g3, b3, w1, g1, b1)
        ctx.s3, ctx.s1, ctx.stride = s3, s1, stride
        return y

    @_commits_sink
    def backward(ctx, dy):
        x, z3, z1, u, w3, g3, b3, w1, g1, b1 = ctx.saved_tensors
        du = H.silu_bwd(u, as_nhwc(dy))
        dz3, dg3, db3 = H.bn_train_bwd(du, z3, g3, b3, ctx.s3, False, dgamma=grad_sink(g3), dbeta=grad_sink(b3))
        dz1, dg1, db1 = H.bn_train_bwd(du, z1, g1, b1, ctx.s1, False, dgamma=grad_sink(g1), dbeta=grad_sink(b1))
        dg3, db3 = (None if grad_sink(g3) is not None else dg3), (None if grad_sink(b3) is not None else db3)
        dg1, db1 = (None if grad_sink(g1) is not None else dg1), (None if grad_sink(b1) is not None else db1)
        if grad_sink(w3) is not None:
            H.conv_wgrad_into(x, dz3, 3, ctx.stride, 1, grad_sink(w3).view(w3.shape[0], 3, 3, w3.shape[1]))
            dw3 = None
        else:
            dw3 = H.conv_wgrad(x, dz3, 3, ctx.stride, 1)
        if grad_sink(w1) is not None:
            H.conv_wgrad_into(x, dz1, 1, ctx.stride, 0, grad_sink(w1).view(w1.shape[0], 1, 1, w1.shape[1]))
            dw1 = None
        else:
            dw1 = H.conv_wgrad(x, dz1, 1, ctx.stride, 0)
        dx = H.conv_dgrad(dz3, H.pack_dgrad(w3, ctx.stride, x.dtype, x.device), ctx.stride)
        if ctx.stride == 2 and w1.shape[2] == 1 and os.environ.get("DYOLO_DGRAD1_SCATTER", "1") != "0":
            # 1x1 stride 2: the gradient reaches the even positions only -- a 1x1 stride-1 convolution at dz's resolution, then a scatter-add
            dx = H.add_dilated2_(dx, H.conv2d(dz1, H.pack_dgrad(w1, 1, x.dtype, x.device, no_accumulate=True)))
        else:
            dx = H.conv_dgrad(dz1, H.pack_dgrad(w1, ctx.stride, x.dtype, x.device), ctx.stride, accumulate=dx)
        return dx, dw3, dg3, db3, dw1, dg1, db1, None, None, None


HEAD_PAD = 16  # class logits are kept in a 16-channel (one fp32 x4 / bf16 x2 chunk multiple) slot of the head buffer


class HeadTail(torch.autograd.Function):
    """cat(conv1x1(xb, wb) + bb, conv1x1(xc, wc) + bc) as ONE fp32 NHWC map per level (Detect's training output,
    head.py:69-72); the class slot is padded to a multiple of 16 channels (pitch = 4*reg_max + pad)."""

    @staticmethod
    def forward(ctx, xb, xc, wb, bb, wc, bc):
        dtype, dev = xb.dtype, xb.device
        n, _, h, w = xb.shape
        nb, nc = wb.shape[0], wc.shape[0]
        ncp = -(-nc // HEAD_PAD) * HEAD_PAD
        buf = H.alloc_nhwc(n, nb + ncp, h, w, torch.float32, dev)
        buf.zero_()
        H.conv2d(xb, H.PackedConv(wb, bb, 1, 0, 1, False, dtype, dev, for_out_f32=True), out=buf[:, :nb], out_f32=True)
        H.conv2d(xc, H.PackedConv(wc, bc, 1, 0, 1, False, dtype, dev, for_out_f32=True), out=buf[:, nb : nb + nc], out_f32=True)
        ctx.save_for_backward(xb, xc, wb, wc)
        ctx.biases = (bb, bc)  # (only their gradient-sink slots are looked up in backward)
        ctx.dims = (nb, nc, ncp)
        return buf[:, : nb + nc]

    @_commits_sink
    def backward(ctx, dy):
        xb, xc, wb, wc = ctx.saved_tensors
        bb, bc = ctx.biases
        nb, nc, ncp = ctx.dims
        dtype, dev = xb.dtype, xb.device
        n, _, h, w = xb.shape
        # one pass: cast, split, zero padding and -- when the gradient comes straight from DetectionLossFn under the trainer -- the seed
        # of the backward pass (LAZY_SEED below), read on the device
        seed = LAZY_SEED.pop(dy.data_ptr(), None)  # by the address DetectionLossFn.backward handed over, before any re-layout
        dzb, dzc = H.head_grad_split(dy if dy.stride(1) == 1 else as_nhwc(dy), nb, nc, ncp, dtype, scale=seed)
        # r04: the 16 Detect-tail parameters go through the trainer's gradient sink like every other parameter (they were the last
        # gradients handed back to autograd: AccumulateGrad nodes created in an eager step and met again under capture on another stream)
        swb, sbb, swc, sbc = grad_sink(wb), grad_sink(bb), grad_sink(wc), grad_sink(bc)
        dwb = H.conv_wgrad(xb, dzb, 1, 1, 0, out=None if swb is None else swb.view(nb, 1, 1, wb.shape[1]))
        dwc = H.conv_wgrad(xc, dzc[:, :nc], 1, 1, 0, out=None if swc is None else swc.view(nc, 1, 1, wc.shape[1]))
        dbb, dbc = H.colsum(dzb, out=sbb), H.colsum(dzc[:, :nc], out=sbc)
        dwb, dwc, dbb, dbc = (None if swb is not None else dwb), (None if swc is not None else dwc), (None if sbb is not None else dbb), (None if sbc is not None else dbc)
        dxb = H.conv_dgrad(dzb, H.pack_dgrad(wb, 1, dtype, dev), 1)
        wcp = torch.zeros((ncp, wc.shape[1], 1, 1), device=dev, dtype=wc.dtype)
        wcp[:nc] = wc.detach()
        dxc = H.conv_dgrad(dzc, H.pack_dgrad(wcp, 1, dtype, dev), 1)
        return dxb, dxc, dwb, dbb, dwc, dbc


class Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return H.upsample2x(x)

    @staticmethod
    def backward(ctx, g):
        return H.upsample2x_bwd(as_nhwc(g))


class ConcatC(torch.autograd.Function):
    """torch.cat(xs, 1) into one NHWC buffer (dy_copy_nhwc); backward hands out channel slices of the gradient."""

    @staticmethod
    def forward(ctx, *xs):
        n, _, h, w = xs[0].shape
        ctx.sizes = [t.shape[1] for t in xs]
        out = H.alloc_nhwc(n, sum(ctx.sizes), h, w, xs[0].dtype, xs[0].device)
        c0 = 0
        for t in xs:
            H.copy_nhwc(t, out[:, c0 : c0 + t.shape[1]])
            c0 += t.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        g = as_nhwc(g)
        outs, c0 = [], 0
        for c in ctx.sizes:
            outs.append(g[:, c0 : c0 + c])
            c0 += c
        return tuple(outs)


class Chunk2(torch.autograd.Function):
    """y.chunk(2, 1) as two channel-slice views (C2f, block.py:239); backward rebuilds one NHWC gradient."""

    @staticmethod
    def forward(ctx, y):
        c = y.shape[1] // 2
        ctx.c = c
        return y[:, :c], y[:, c:]

    @staticmethod
    def backward(ctx, ga, gb):
        ref = ga if ga is not None else gb
        n, _, h, w = ref.shape
        out = H.alloc_nhwc(n, 2 * ctx.c, h, w, ref.dtype, ref.device)
        for i, g in enumerate((ga, gb)):
            sl = out[:, i * ctx.c : (i + 1) * ctx.c]
            if g is None:
                sl.zero_()
            else:
                H.copy_nhwc(as_nhwc(g), sl)
        return out


class AddT(torch.autograd.Function):
    """a + b (Bottleneck shortcut) through dy_add_nhwc."""

    @staticmethod
    def forward(ctx, a, b):
        return H.add_nhwc(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


class SppfPool(torch.autograd.Function):
    """cat(x, m(x), m(m(x)), m(m(m(x)))) with m = MaxPool2d(k, 1, k//2) (SPPF, block.py:187-191)."""

    @staticmethod
    def forward(ctx, x, k):
        n, c, h, w = x.shape
        buf = H.alloc_nhwc(n, 4 * c, h, w, x.dtype, x.device)
        H.copy_nhwc(x, buf[:, :c])
        H.sppf_maxpool3(buf[:, :c], buf[:, c : 2 * c], buf[:, 2 * c : 3 * c], buf[:, 3 * c :], k)
        ctx.save_for_backward(buf)
        ctx.k, ctx.c = k, c
        return buf

    @staticmethod
    def backward(ctx, g):
        (buf,) = ctx.saved_tensors
        c, k = ctx.c, ctx.k
        g0 = as_nhwc(g)
        g = H.alloc_nhwc(g0.shape[0], g0.shape[1], g0.shape[2], g0.shape[3], g0.dtype, g0.device)  # private copy: accumulated in place
        H.copy_nhwc(g0, g)
        s = lambda t, i: t[:, i * c : (i + 1) * c]  # noqa: E731
        H.maxpool_bwd(s(buf, 2), s(g, 3), s(g, 2), k, True)  # g2 += bwd(y2 -> y3, g3)
        H.maxpool_bwd(s(buf, 1), s(g, 2), s(g, 1), k, True)  # g1 += bwd(y1 -> y2, g2)
        H.maxpool_bwd(s(buf, 0), s(g, 1), s(g, 0), k, True)  # gx += bwd(x -> y1, g1)
        return s(g, 0), None


class DetectionLossFn(torch.autograd.Function):
    """v8DetectionLoss on the per-level fp32 head maps through dy_detection_loss (value + gradient in one call)."""

    @staticmethod
    def forward(ctx, gt, strides, nc, reg_max, hyp, *levels):
        out, _, grads = H.detection_loss(levels, gt, strides, nc, reg_max, box=hyp[0], cls=hyp[1], dfl=hyp[2], want_grad=True)
        ctx.grads = grads
        ctx.nch = levels[0].shape[1]
        total, items = out[3].clone(), out[:3].clone()
        ctx.mark_non_differentiable(items)
        return total, items

    @staticmethod
    def backward(ctx, g_total, g_items):
        if g_total is None:
            return (None, None, None, None, None, *([None] * len(ctx.grads)))
        gs = [g[:, : ctx.nch] for g in ctx.grads]
        if _LAZY[0] and g_total.is_cuda and g_total.dtype == torch.float32 and g_total.numel() == 1:
            # the trainer's own backward: d total / d level is handed on UNSCALED and HeadTail.backward multiplies by the seed inside its
            # split kernel (a device read) -- a full-map torch multiply per level otherwise
            for g in gs:
                LAZY_SEED[g.data_ptr()] = g_total
            return (None, None, None, None, None, *gs)
        return (None, None, None, None, None, *[g * g_total for g in gs])
