"""Tensor-level wrappers over the libdyolo C-ABI.

Activations are torch tensors that are *logically* (N, C, H, W) — the shape the reference's
modules exchange (nn/tasks.py:148-154) — but live in NHWC memory: ``stride(1) == 1`` and
``stride(3)`` is the pixel pitch ``ld``.  A channel slice ``buf[:, c0:c1]`` of such a tensor is a
valid view for every kernel, which is how Concat / chunk are done without copies.

PyTorch is used here for device memory and streams only; every arithmetic op on the hot path
is a libdyolo kernel.  A ``LaunchPlan`` records the (function, arguments) list of a forward so
later forwards with the same shapes replay the launches without re-running the Python modules.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import os
import warnings

import torch

from . import _lib
from ._lib import DY_ACT_NONE, DY_ACT_SILU, DY_ACT_SILU_L2E, BnDesc, BranchDesc, C2fDesc, ConvDesc, DecodeDesc, HeadDecodeDesc, LossDesc, NmsDesc, Stem2Desc, check, lib

FP8 = torch.float8_e4m3fn  # OCP e4m3fn: gfx950's fp8 (MI300's fnuz is another encoding)
# DY_F16X2 (include/dyolo.h): split float16 pairs, x ~= hi + lo * 2^-11 — the bar-exact precision on the 16-bit MFMA.  torch has no such
# type; its tensors are carried as torch.complex32 — a 4-byte element of two halves — purely as a CONTAINER: shapes, strides, channel
# slices at multiples of 8 and raw copies mean what they say, arithmetic on them in torch does not (inside a pixel row the halves are
# laid out [hi x 8 | lo x 8] per 8 channels, not interleaved per element).  ``to_nchw_f32`` gives the values back.
F16X2 = torch.complex32
_DTYPES = {torch.bfloat16: _lib.DY_BF16, torch.float16: _lib.DY_F16, torch.float32: _lib.DY_F32, FP8: _lib.DY_FP8, F16X2: _lib.DY_F16X2}
warnings.filterwarnings("ignore", message="ComplexHalf support is experimental")

# DY_FP8: ONE activation scale for the whole network (real value = quantum * scale), per-output-channel weight scales
# (include/dyolo.h, dy_conv_desc.w_scale / act_scale).  Set by the predictor from a calibration pass before any fp8 pack is built.
_FP8 = {"act_scale": 1.0}


def set_fp8_act_scale(scale: float) -> None:
    if not scale > 0:
        raise ValueError("fp8 activation scale must be positive")
    _FP8["act_scale"] = float(scale)


def fp8_act_scale() -> float:
    return _FP8["act_scale"]


_absmax_log: Optional[list] = None  # calibration: conv2d appends the absolute maximum of every output while this is a list


class observe_absmax:
    """Context manager: records max |y| of every conv output launched inside (device scalars; read them after a sync)."""

    def __enter__(self):
        global _absmax_log
        _absmax_log = []
        return _absmax_log

    def __exit__(self, *exc):
        global _absmax_log
        _absmax_log = None
        return False



def dy_dtype(dt: torch.dtype) -> int:
    try:
        return _DTYPES[dt]
    except KeyError:
        raise TypeError(f"unsupported activation dtype {dt}; use bfloat16, float16 or float32") from None


_ESIZE = {torch.bfloat16: 2, torch.float16: 2, torch.float32: 4, FP8: 1, F16X2: 4}


def elems_per_chunk(dt: torch.dtype) -> int:
    return 16 // (_ESIZE.get(dt) or torch.empty((), dtype=dt).element_size())


def chan_gran(dt: torch.dtype) -> int:
    """Channels per addressable group of a pixel row: one 16-byte chunk, or — split float16 — a (hi, lo) chunk pair of 8 channels."""
    return 8 if dt == F16X2 else elems_per_chunk(dt)


def require_device(t: torch.Tensor, what: str = "tensor") -> None:
    """The hot path is HIP only: refuse CPU tensors loudly instead of falling back."""
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on '{t.device}': the Drone-YOLO hot path runs on an MI355X (HIP) device only; "
            "there is no CPU fallback in this package."
        )


def alloc_nhwc(n: int, c: int, h: int, w: int, dtype: torch.dtype, device, ld: Optional[int] = None) -> torch.Tensor:
    """Logical (n,c,h,w) tensor in NHWC memory with pixel pitch ``ld`` (default c)."""
    ld = c if ld is None else ld
    buf = torch.empty((n, h, w, ld), dtype=dtype, device=device)
    return buf.permute(0, 3, 1, 2)[:, :c]


def view_params(t: torch.Tensor) -> Tuple[int, int]:
    """(data_ptr, ld) of an NHWC view; raises if ``t`` is not one."""
    if t.dim() != 4:
        raise ValueError(f"expected a 4-D (N,C,H,W) tensor, got shape {tuple(t.shape)}")
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
    ok = (c == 1 or sc == 1) and (w == 1 or sw == ld) and (h == 1 or sh == w * ld) and (n == 1 or sn == h * w * ld)
    if not ok or ld < c:
        raise ValueError(
            f"tensor of shape {tuple(t.shape)} / strides {t.stride()} is not an NHWC view; "
            "use hip_ops.to_nhwc() or torch.channels_last"
        )
    return t.data_ptr(), ld


# ---- launch plan ------------------------------------------------------------------------------


class LaunchPlan:
    """Recorded kernel launches of one forward: list of (cfunc, args) + the buffers they use."""

    def __init__(self):
        self.ops: List[Tuple[object, tuple, bool]] = []  # (cfunc, args-without-stream, by_desc)
        self.tags: List[object] = []  # per op: the model layer that issued it (``layer_tag``; None outside a layer) — per-layer profiling
        self.keep: List[object] = []  # tensors / descriptors that must outlive the plan
        self.input_slot: Optional[Tuple[int, int]] = None  # (op index, arg index) of the user input pointer
        self.outputs = None

    def replay(self, stream: int, start: int = 0, stop: Optional[int] = None) -> None:
        """Issue the recorded launches (ops[start:stop]) on ``stream``."""
        for fn, args, _ in (self.ops if start == 0 and stop is None else self.ops[start:stop]):
            rc = fn(*args, stream)
            if rc:
                check(rc, fn.__name__)

    def rebind_input(self, ptr: int) -> None:
        i, j = self.input_slot
        fn, args, d = self.ops[i]
        if j < 0:  # the input pointer is field ``x`` of the descriptor passed by reference
            args[0]._obj.x = ptr
            return
        args = list(args)
        args[j] = ptr
        self.ops[i] = (fn, tuple(args), d)


_recording: Optional[LaunchPlan] = None
_LAYER_TAG = [None]  # (index, type name) of the model layer whose forward is running (set by nn/tasks.py::_predict_layers)


class layer_tag:
    """Launches issued inside are attributed to model layer ``tag`` in a recording plan (``LaunchPlan.tags``)."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        self.prev, _LAYER_TAG[0] = _LAYER_TAG[0], self.tag

    def __exit__(self, *exc):
        _LAYER_TAG[0] = self.prev
        return False


class record:
    """Context manager: launches issued inside are also appended to ``plan``."""

    def __init__(self, plan: LaunchPlan):
        self.plan = plan

    def __enter__(self):
        global _recording
        if _recording is not None:
            raise RuntimeError("nested LaunchPlan recording")
        _recording = self.plan
        return self.plan

    def __exit__(self, *exc):
        global _recording
        _recording = None
        return False


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _launch(fn, args: tuple, keep: Sequence[object] = (), record: bool = True) -> None:
    """``record=False``: one-time preparation (weight packing) that must not be replayed with a plan recorded around it."""
    rc = fn(*args, _stream())
    if rc:
        check(rc, fn.__name__)
    if _recording is not None and record:
        _recording.ops.append((fn, args, False))
        _recording.tags.append(_LAYER_TAG[0])
        _recording.keep.extend(keep)


# ---- convolution --------------------------------------------------------------------------------


def _pad_to(t: torch.Tensor, shape) -> torch.Tensor:
    """``t`` (fp32) zero-padded at the end of every dim to ``shape``; ``t`` itself when nothing is to pad (no kernel)."""
    if tuple(t.shape) == tuple(shape):
        return t
    out = torch.zeros(shape, dtype=t.dtype, device=t.device)
    out[tuple(slice(0, n) for n in t.shape)] = t
    return out


def _permute_cast(view: torch.Tensor, dtype: torch.dtype, device) -> torch.Tensor:
    """Contiguous copy of a permuted view in ``dtype`` — one copy kernel for layout change + cast."""
    out = torch.empty(view.shape, dtype=dtype, device=view.device)
    out.copy_(view)
    return out.view(-1).to(device)


_ZERO_BIAS = {}


def zero_bias(n: int, device) -> torch.Tensor:
    """Cached fp32 zeros(n) on ``device`` (BatchNorm-ed and gradient convolutions carry no bias); never written to."""
    key = (n, str(device))
    z = _ZERO_BIAS.get(key)
    if z is None:
        z = _ZERO_BIAS[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return z


# ---- the log2(e)-scaled activation domain (DY_ACT_SILU_L2E, include/dyolo.h) ---------------------------------------------
LOG2E = 1.4426950408889634
_SCALED = [False]


class scaled_activations:
    """While active, weights are packed for a pass whose stored activations are log2(e) times the reference's: every bias times
    log2(e), the weights of a layer that reads raw data (the image) likewise, SiLU as DY_ACT_SILU_L2E (one VALU instruction less per
    output element), the plain 1x1 convolutions that end the Detect branches divided by log2(e) so that the logits come out in true
    units.  Max pool, nearest upsample, Concat and the Bottleneck sum commute with the positive scale.  The model executor
    (nn/tasks.py::_predict_once) turns it on for 16-bit / fp8 storage; modules called on their own stay in the reference's units."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _SCALED[0]
        _SCALED[0] = self.on
        return self

    def __exit__(self, *exc):
        _SCALED[0] = self.prev
        return False


def scaled_domain() -> bool:
    return _SCALED[0]


def domain_fold(w: torch.Tensor, b: torch.Tensor, silu: bool, raw_input: bool = False, raw_output: bool = False):
    """(weight, bias, activation code) of one convolution for the activation domain in force (see ``scaled_activations``)."""
    if not _SCALED[0]:
        return w, b, (DY_ACT_SILU if silu else DY_ACT_NONE)
    w, b = w.detach().float(), b.detach().float()
    if raw_output:
        if silu or raw_input:
            raise ValueError("domain_fold: a layer that leaves the scaled domain is a plain convolution on scaled input")
        return w / LOG2E, b, DY_ACT_NONE
    return (w * LOG2E if raw_input else w), b * LOG2E, (DY_ACT_SILU_L2E if silu else DY_ACT_NONE)


def _act_code(act) -> int:
    return int(act) if isinstance(act, int) and not isinstance(act, bool) else (DY_ACT_SILU if act else DY_ACT_NONE)


class PackCache:
    """Weight packing of a whole training step in one launch.  A step packs every convolution's fp32 master weights twice (forward
    and input-gradient form): ~160 ``dy_pack_conv_weights`` launches of ~5 us.  The first step under ``batched_weight_packing`` packs
    one by one and remembers each job with a PERSISTENT destination; from then on ``pack_all()`` (the trainer calls it at the start of
    a step, i.e. after the optimizer changed the weights) runs all of them as ONE ``dy_pack_conv_weights_batched`` launch and the
    ``PackedConv`` constructors of that step take the packed buffers as they are.  Sources are the flat parameter buffer's views
    (stable addresses); a job that first appears later is packed on its own and joins the table at the next ``pack_all`` outside a
    stream capture (the table is uploaded with a host-to-device copy)."""

    def __init__(self, dtype: torch.dtype, device, stable: torch.Tensor):
        """``stable``: the buffer whose views may be cached (the trainer's flat parameter buffer) -- weights built per step
        (a padded copy, a folded RepVGG kernel) have another address every time and are packed on their own."""
        self.dtype, self.device = dtype, torch.device(device)
        self.lo, self.hi = stable.data_ptr(), stable.data_ptr() + stable.numel() * stable.element_size()
        self.jobs: dict = {}      # job tuple -> [destination, generation packed, keep-alive source]
        self.table = None         # (device table, n_jobs, total_blocks, job order)
        self.dirty = False
        self.gen = 0
        self._retired = []  # superseded job tables (see _build)

    def lookup(self, job):
        e = self.jobs.get(job)
        return e[0] if e is not None and e[1] == self.gen else None

    def slot(self, job, src) -> torch.Tensor:
        e = self.jobs.get(job)
        if e is None:
            e = self.jobs[job] = [torch.empty(job[11], dtype=self.dtype, device=self.device), -1, src]
            self.dirty = True
        e[1] = self.gen  # the caller packs it now
        return e[0]

    def _build(self) -> None:
        L = lib()
        order = list(self.jobs)
        n = len(order)
        arr = (_lib.PackJob * n)()
        for i, j in enumerate(order):
            a = arr[i]
            a.w, a.s_co, a.s_ci, a.s_r, a.s_q, a.cout, a.cin, a.ksize, a.transpose_flip, a.cin_logical, a.w_layout = j[:11]
            a.dst, a.dst_elems = self.jobs[j][0].data_ptr(), j[11]
        nbytes = int(L.dy_pack_conv_weights_table_bytes(n))
        host = torch.empty(nbytes, dtype=torch.uint8)
        blocks = C.c_int32(0)
        check(L.dy_pack_conv_weights_table(arr, n, dy_dtype(self.dtype), host.data_ptr(), nbytes, C.byref(blocks)), "dy_pack_conv_weights_table")
        if self.table is not None:
            # a captured step graph has the old table's device address baked into its dy_pack_conv_weights_batched node: a table that
            # was ever launched stays allocated for the life of the cache (a few KB each; rebuilt only when a new layer shape appears)
            self._retired.append(self.table[0])
        self.table = (host.to(self.device), n, int(blocks.value), order)
        self.dirty = False

    def pack_all(self) -> None:
        """Start of a step: everything known is packed again from the current weights."""
        if not self.jobs:
            return
        if (self.dirty or self.table is None) and not torch.cuda.is_current_stream_capturing():
            self._build()
        self.gen += 1
        if self.table is None:
            return
        tab, n, blocks, order = self.table
        _launch(lib().dy_pack_conv_weights_batched, (tab.data_ptr(), n, blocks, dy_dtype(self.dtype)), record=False)
        for j in order:
            self.jobs[j][1] = self.gen


_PACK_CACHE = {"active": None}
HREG_128 = [os.environ.get("DYOLO_HREG_128", "1") != "0"]  # see PackedConv: 128-channel 3x3 layers on the register-weight kernel (0: the virtual-flat GEMM, for A/B runs)
FLAT_K_3X3 = [os.environ.get("DYOLO_FLAT_K_3X3", "1") != "0"]  # see PackedConv: odd-width 3x3 layers on the flat-K kernel (0: the halo kernel, for A/B runs)


@contextlib.contextmanager
def batched_weight_packing(cache: Optional[PackCache]):
    """``PackedConv`` constructions of device-resident fp32 weights inside this block go through ``cache`` (see PackCache)."""
    prev = _PACK_CACHE["active"]
    _PACK_CACHE["active"] = cache
    try:
        yield cache
    finally:
        _PACK_CACHE["active"] = prev


class PackedConv:
    """Folded + packed weights of one convolution in the layout ``dy_conv2d_nhwc`` expects.

    ``weight``: (cout, cin/groups, k, k) fp32 with BatchNorm / RepVGG branches already folded,
    ``bias``: (cout,) fp32.  Packing = (cout, k, k, cin) row-major, rows padded to k_pad, rows
    count padded to cout_pad with zeros (include/dyolo.h).
    """

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, stride: int, pad: int, groups: int, act: bool,
                 dtype: torch.dtype, device, cin_pad: Optional[int] = None, halo: Optional[bool] = None,
                 for_out_f32: bool = False, transpose_flip: bool = False):
        """``halo`` = False forces the generic row layout (both special layouts off); ``for_out_f32``: the conv
        will be called with out_f32=True (Detect heads), which narrows the shapes the streaming kernel is built for.
        ``transpose_flip``: pack W'[ci][co][r][q] = weight[co][ci][k-1-r][k-1-q] (the input-gradient convolution) — device packing only.
        fp32 weights that already live on the device (training: the master weights, every step) are packed by ONE
        ``dy_pack_conv_weights`` launch instead of the pad / permute / flip / cast chain below."""
        L = lib()
        # what a DY_WLAYOUT_ROWS pack of the same layer is built from (``rows()``: calls the special layout's kernels refuse)
        self._rows_args = (weight, bias, stride, pad, groups, act, dtype, device, cin_pad, for_out_f32, transpose_flip)
        self._rows_pack = None
        dev_pack = weight.is_cuda and weight.dtype == torch.float32 and dtype not in (FP8, F16X2) and groups == 1 and weight.device == torch.device(device)
        if transpose_flip and not dev_pack:
            raise ValueError("PackedConv(transpose_flip=True) needs fp32 weights on the target device")
        self._src = weight if dev_pack else None
        cin_logical = 0
        if dev_pack:
            src = weight.detach()
            if cin_pad is not None and cin_pad > weight.shape[1]:
                cin_logical = cin_pad
            lshape = (src.shape[1], src.shape[0], src.shape[2], src.shape[3]) if transpose_flip else tuple(src.shape)
            if cin_logical:
                lshape = (lshape[0], cin_logical, lshape[2], lshape[3])
            weight = torch.empty(lshape, device="meta")  # shape carrier for the layout decision below; data moves in _device_pack

            def _device_pack(layout: int, n_elems: int) -> torch.Tensor:
                st = src.stride()
                job = (src.data_ptr(), st[0], st[1], st[2], st[3], src.shape[0], src.shape[1], src.shape[2], int(transpose_flip), cin_logical, layout, n_elems)
                cache = _PACK_CACHE["active"]
                if cache is not None and cache.dtype == dtype and cache.lo <= job[0] < cache.hi:
                    hit = cache.lookup(job)
                    if hit is not None:  # packed by this step's dy_pack_conv_weights_batched launch
                        return hit
                    out = cache.slot(job, src)
                else:
                    out = torch.empty(n_elems, dtype=dtype, device=src.device)
                _launch(L.dy_pack_conv_weights, job[:10] + (out.data_ptr(), n_elems, dy_dtype(dtype), layout), record=False)
                return out

            def _device_bias(n: int) -> torch.Tensor:
                if bias.numel() == n and bias.is_cuda and bias.dtype == torch.float32:
                    return bias.detach()
                if bias is _ZERO_BIAS.get((bias.numel(), str(src.device))):  # the shared zeros (training, gradient convolutions): no fill + copy per call
                    return zero_bias(n, src.device)
                bp = torch.zeros((n,), dtype=torch.float32, device=src.device)
                bp[: bias.numel()] = bias.detach().to(torch.float32)
                return bp
        elif cin_pad is not None and cin_pad > weight.shape[1]:
            # the input view carries zero-padded channels (the 3-channel image padded to one 16-byte chunk)
            if groups != 1:
                raise ValueError("cin_pad is only meaningful for dense convolutions")
            weight = torch.nn.functional.pad(weight.detach().float(), (0, 0, 0, 0, 0, cin_pad - weight.shape[1]))
        cout, cin_g, k, k2 = weight.shape
        wdev = weight.device  # pack where the weights live (CPU for inference setup, the GPU in training: no host round trip)
        assert k == k2, "square kernels only"
        self.cout, self.cin, self.k, self.stride, self.pad, self.groups = cout, cin_g * groups, k, stride, pad, groups
        self.act = _act_code(act)  # bool (SiLU or none) or a dy_act code
        self.dtype = dtype
        self.layout = _lib.DY_WLAYOUT_ROWS
        self.wscale, self.act_scale = None, 1.0
        if dtype == F16X2:
            # split float16 (include/dyolo.h, DY_F16X2): every output-channel row times the power of two that puts its largest weight in
            # [2^13, 2^14) (exact; the inverse goes to w_scale and multiplies the accumulator), then hi = rn_f16(w) — 0 below float16's
            # smallest normal — and lo = rn_f16(w - hi), UNSCALED: the kernel multiplies x_hi by it directly and forms w_hi 2^-11 for the
            # x_lo term itself.  Row layout: K order (r, q, c), every 8 channels as [hi x 8 | lo x 8]: 4 bytes per element.
            if groups != 1:
                # DWConv of the -sf YAML: the small-group kernel multiplies joined fp32 inputs by fp32 weights (rows [cout][taps][cin / groups])
                if cout != groups or cin_g not in (1, 2, 4) or cout % 8:
                    raise NotImplementedError("split-float16 storage of a grouped convolution: one output channel per group, 1 / 2 / 4 inputs each, cout % 8 == 0")
                self.w = weight.detach().to(torch.float32).permute(0, 2, 3, 1).reshape(cout, k * k * cin_g).contiguous().to(device)
                self.b = bias.detach().to(torch.float32).contiguous().to(device)
                self.k_pad, self.cout_pad = k * k * cin_g, cout
                self._rows_args = None
                return
            w = weight.detach().to(torch.float32).permute(0, 2, 3, 1).reshape(cout, k * k, cin_g)
            if cin_g % 8:
                w = torch.nn.functional.pad(w, (0, 8 - cin_g % 8))
            self.cin = w.shape[2]
            amax = w.abs().amax((1, 2)).clamp_min(1e-30)
            sc = torch.exp2(13.0 - torch.floor(torch.log2(amax)))
            ws = w * sc[:, None, None]
            hi = torch.where(ws.abs() < 2.0 ** -14, torch.zeros_like(ws), ws).to(torch.float16)
            lo = (ws - hi.to(torch.float32)).to(torch.float16)
            self.k_pad, self.cout_pad = L.dy_conv_k_pad(self.cin, k, dy_dtype(dtype)), L.dy_conv_cout_pad(cout)
            pair = torch.stack((hi.reshape(cout, k * k, self.cin // 8, 8), lo.reshape(cout, k * k, self.cin // 8, 8)), 3)  # (cout, taps, groups, 2, 8)
            wp = torch.zeros((self.cout_pad, self.k_pad * 2), dtype=torch.float16, device=wdev)
            wp[:cout, : k * k * self.cin * 2] = pair.reshape(cout, -1)
            bp = torch.zeros((self.cout_pad,), dtype=torch.float32, device=wdev)
            bp[:cout] = bias.detach().to(torch.float32).to(wdev)
            sp = torch.ones((self.cout_pad,), dtype=torch.float32, device=wdev)
            sp[:cout] = 1.0 / sc
            self.w, self.b, self.wscale = wp.contiguous().to(device), bp.contiguous().to(device), sp.contiguous().to(device)
            self._rows_args = None
            return
        if dtype == FP8:
            # e4m3 weights with one scale per OUTPUT channel (absmax -> 448), rows layout only; the epilogue multiplies the
            # accumulator by act_scale * weight_scale[co] (include/dyolo.h)
            if groups != 1:
                raise NotImplementedError("fp8 storage is built for dense convolutions")
            self.act_scale = fp8_act_scale()
            w = weight.detach().to(torch.float32).permute(0, 2, 3, 1).reshape(cout, k * k * cin_g)
            ws = (w.abs().amax(1) / 448.0).clamp_min(1e-12)
            self.k_pad, self.cout_pad = L.dy_conv_k_pad(self.cin, k, dy_dtype(dtype)), L.dy_conv_cout_pad(cout)
            wp = torch.zeros((self.cout_pad, self.k_pad), dtype=torch.float32, device=wdev)
            wp[:cout, : w.shape[1]] = w / ws[:, None]
            bp = torch.zeros((self.cout_pad,), dtype=torch.float32, device=wdev)
            bp[:cout] = bias.detach().to(torch.float32).to(wdev)
            sp = torch.ones((self.cout_pad,), dtype=torch.float32, device=wdev)
            sp[:cout] = ws * self.act_scale
            self.w = wp.to(FP8).contiguous().to(device)
            self.b, self.wscale = bp.contiguous().to(device), sp.contiguous().to(device)
            self._rows_args = None
            return
        # stride-2 halo tiles are 17x33 pixels (2 x 45 KB of LDS): only a weight set of <= 36 KB fits beside them
        s2_fits = self.cin <= 4 * elems_per_chunk(dtype) and cout > 32 or self.cin <= 8 * elems_per_chunk(dtype) and cout <= 32
        # r03: the 64-channel downsampling layers run on the register-weight stride-2 kernel (conv3x3_hreg_s2), same fragment layout
        s2_fits = s2_fits or (dtype in (torch.bfloat16, torch.float16) and self.cin == 64 and cout % 64 == 0 and cout <= 256 and halo is not False)
        # deep 3x3 layers (small maps, weight sets far beyond LDS) run faster as a flat-M implicit GEMM on the LDS-DMA
        # big-tile kernel behind DY_WLAYOUT_ROWS (conv_gemm_glds.hip): measured at batch 128, 256->256 @20x20 90 vs 162 us
        kstep = 8 * elems_per_chunk(dtype)
        # r04: 128 input channels fit the register-weight kernel too (four 32-channel chunks: 144 weight registers, two workgroups per CU):
        # 128->128 @40x40 149 -> 136 us, @80x80 543 -> 437 us (1,100 TFLOP/s) at B = 256 against the virtual-flat GEMM
        # r05: bf16 INFERENCE packs stay on the virtual-flat GEMM: the four-chunk summation order moved the s640 fixture's worst bf16 box from
        # IoU 0.9981 to 0.9979, under the 0.998 floor of tests/test_model_gpu.py, and the floor is not what gives way; fp16 -- the
        # 16-bit inference dtype, 8x finer -- did not move, and the raw convolutions of a training step (no activation: BatchNorm follows) keep
        # the kernel for its statistics epilogue
        hreg128 = HREG_128[0] and halo is None and k == 3 and stride == 1 and pad == 1 and groups == 1 and \
            (dtype == torch.float16 or (dtype == torch.bfloat16 and _act_code(act) == 0)) and self.cin == 128 and cout % 64 == 0 and cout <= 256
        deep3x3 = halo is None and k == 3 and stride == 1 and self.cin % kstep == 0 and cout % 64 == 0 and \
            self.cin >= 128 and cout >= 128 and not hreg128
        # (r04: deep layers whose cout is no multiple of 128 — scale x: 320 -> 320 — take the flat-K kernel's 160-wide tiles behind the same
        # layout (csrc/conv_igemm.hip: fk_first): 19.09 -> 18.68 ms per x1536 pass against the halo kernel, 536 -> 860 TFLOP/s against the
        # 64-cout persistent tiles of the tap-aligned kernel)
        # r04: 3x3 layers whose channel counts are not whole 64-channel K-steps / 64-cout tiles (the n / m / x scales: 80, 160, 320 ...)
        # run on the flat-K LDS-DMA kernel behind DY_WLAYOUT_ROWS (conv_gemm_fk.hip) instead of the halo kernel's 32-channel chunks
        flat3x3 = FLAT_K_3X3[0] and halo is None and k == 3 and pad == 1 and groups == 1 and dtype in (torch.bfloat16, torch.float16) and \
            self.cin >= 64 and (self.cin % 64 != 0 or cout % 64 != 0) and self.cin % 8 == 0 and cout % 8 == 0
        if (halo is None or halo) and not deep3x3 and not flat3x3 and groups == 1 and k == 3 and pad == 1 and (stride == 1 or (stride == 2 and s2_fits)) \
                and cout % 4 == 0 and self.cin >= 4 * elems_per_chunk(dtype) // 2:
            # LDS-halo 3x3 kernel: MFMA-fragment-ordered weights (include/dyolo.h, DY_WLAYOUT_HALO3X3)
            self.layout = _lib.DY_WLAYOUT_HALO3X3
            e = elems_per_chunk(dtype)
            kc, bn = 4 * e, (64 if cout > 32 else 32)
            nt, nch = -(-cout // bn), -(-self.cin // kc)
            self.k_pad, self.cout_pad = 0, L.dy_conv_cout_pad(cout)
            if dev_pack:
                self.w, self.b = _device_pack(self.layout, nt * nch * 9 * (bn // 16) * 64 * e), _device_bias(self.cout_pad)
                return
            wpad = _pad_to(weight.detach().to(torch.float32), (nt * bn, nch * kc, 3, 3))
            self.w = _permute_cast(wpad.reshape(nt, bn // 16, 16, nch, 4, e, 3, 3).permute(0, 3, 6, 7, 1, 4, 2, 5), dtype, device)
            self.b = _pad_to(bias.detach().to(torch.float32).to(wdev), (self.cout_pad,)).to(device)
            return
        e = elems_per_chunk(dtype)
        nkg = -(-self.cin // (4 * e))
        frag_ok = nkg in (2, 3, 4, 6, 8, 12) and (cout > 16 or nkg == 2) and self.cin % e == 0
        if for_out_f32 and dtype != torch.float32:
            frag_ok = frag_ok and nkg == 2 and cout <= 64
        if frag_ok:  # the streaming kernel's LDS footprint (conv1x1_stream.hip::launch_1x1) must fit 160 KB
            bn_ = 128 if cout > 64 else (64 if cout > 16 else 16)
            osz = 4 if (for_out_f32 or dtype == torch.float32) else 2
            smem = nkg * (bn_ // 16) * 1024 + bn_ * 4 + 8 * (2 if nkg <= 2 else 1) * 16 * (bn_ * osz + 16)
            frag_ok = smem <= 160 * 1024
        if (halo is None or halo) and groups == 1 and k == 1 and stride == 1 and pad == 0 and frag_ok:
            # streaming 1x1 kernel: fragment-ordered weights, single tap (include/dyolo.h, DY_WLAYOUT_FRAG1X1)
            self.layout = _lib.DY_WLAYOUT_FRAG1X1
            kc, bn = 4 * e, (128 if cout > 64 else (64 if cout > 16 else 16))
            nt = -(-cout // bn)
            if dev_pack:
                self.k_pad, self.cout_pad = 0, max(L.dy_conv_cout_pad(cout), nt * bn)
                self.w, self.b = _device_pack(self.layout, nt * nkg * (bn // 16) * 64 * e), _device_bias(self.cout_pad)
                return
            wpad = torch.zeros((nt * bn, nkg * kc), dtype=torch.float32, device=wdev)
            wpad[:cout, : self.cin] = weight.detach().to(torch.float32).view(cout, self.cin)
            wp = wpad.view(nt, bn // 16, 16, nkg, 4, e).permute(0, 3, 1, 4, 2, 5).contiguous().view(-1)
            self.k_pad, self.cout_pad = 0, max(L.dy_conv_cout_pad(cout), nt * bn)
            bp = torch.zeros((self.cout_pad,), dtype=torch.float32, device=wdev)
            bp[:cout] = bias.detach().to(torch.float32).to(wdev)
            self.w = wp.to(dtype).contiguous().to(device)
            self.b = bp.contiguous().to(device)
            return
        if groups == 1:
            self.k_pad = L.dy_conv_k_pad(self.cin, k, dy_dtype(dtype))
            self.cout_pad = L.dy_conv_cout_pad(cout)
            if dev_pack:
                self.w, self.b = _device_pack(self.layout, self.cout_pad * self.k_pad), _device_bias(self.cout_pad)
                return
            if self.k_pad == k * k * cin_g and self.cout_pad == cout:  # nothing to pad: one permute + cast kernel
                self.w = _permute_cast(weight.detach().permute(0, 2, 3, 1), dtype, device)
                self.b = bias.detach().to(torch.float32).to(device)
                return
            w = weight.detach().to(torch.float32).permute(0, 2, 3, 1).reshape(cout, k * k * cin_g)
            wp = torch.zeros((self.cout_pad, self.k_pad), dtype=torch.float32, device=wdev)
            wp[:cout, : w.shape[1]] = w
            bp = torch.zeros((self.cout_pad,), dtype=torch.float32, device=wdev)
            bp[:cout] = bias.detach().to(torch.float32).to(wdev)
        else:
            w = weight.detach().to(torch.float32).permute(0, 2, 3, 1).reshape(cout, k * k * cin_g)
            self.k_pad, self.cout_pad = w.shape[1], cout
            wp, bp = w, bias.detach().to(torch.float32)
        self.w = wp.to(dtype).contiguous().to(device)
        self.b = bp.contiguous().to(device)

    def rows(self) -> "PackedConv":
        """The same layer packed in DY_WLAYOUT_ROWS (built on first use, kept): the layout every call shape has a kernel for.
        ``conv2d`` takes it when a call does not meet the preconditions of the kernels behind this pack's special layout."""
        if self.layout == _lib.DY_WLAYOUT_ROWS:
            return self
        if self._rows_pack is None:
            w, b, stride, pad, groups, act, dtype, device, cin_pad, for_out_f32, tf = self._rows_args
            self._rows_pack = PackedConv(w, b, stride, pad, groups, act, dtype, device, cin_pad=cin_pad, halo=False, for_out_f32=for_out_f32, transpose_flip=tf)
        return self._rows_pack

    def for_call(self, x_bytes: int, y_bytes: int, ld_y: int, y_ptr: int, residual: bool, out_f32: bool, gathered: bool) -> "PackedConv":
        """This pack, or its DY_WLAYOUT_ROWS twin when the call is outside what the special layout's kernels take (mirrors
        csrc/conv3x3_halo.hip::conv3x3_halo_dispatch and conv3x3_hreg.hip::conv3x3_hreg_try): a second source / upsampled or dilated
        gather, an input view beyond the 4 GiB a buffer descriptor addresses, and -- stride-2 64-channel layers, which have no kernel
        but conv3x3_hreg_s2 behind DY_WLAYOUT_HALO3X3 -- a residual, fp32 output, an output pitch / base that rules out 16-byte
        stores or an input view beyond 2 GiB."""
        if self.layout != _lib.DY_WLAYOUT_HALO3X3:
            return self
        if gathered or x_bytes >= (1 << 32) - 64:
            return self.rows()
        if self.stride == 2 and self.cin == 64 and self.cout > 32:
            if residual or out_f32 or ld_y % 8 or y_ptr % 16 or x_bytes >= (1 << 31) or y_bytes >= (1 << 32) - 64 or self.dtype == torch.float32:
                return self.rows()
        if self.stride == 1 and self.cin == 128:  # packed for conv3x3_hreg's four-chunk form: what it declines runs on the virtual-flat GEMM
            if residual or out_f32 or ld_y % 8 or y_ptr % 16 or x_bytes >= (1 << 31) or y_bytes >= (1 << 32) - 64:
                return self.rows()
        return self


def conv_out_hw(h: int, w: int, k: int, s: int, p: int) -> Tuple[int, int]:
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


def last_kernel_name() -> str:
    """Device kernel the calling thread's last libdyolo launch dispatched to (``dy_last_kernel_name``)."""
    return (lib().dy_last_kernel_name() or b"").decode()


class BnBehind:
    """The train-mode BatchNorm + activation of the layer IN FRONT of a convolution, for that convolution's input gradient
    (``conv_dgrad(bn_behind=)``, ``dy_conv_desc.bnb_z``): when the gradient kernel has the epilogue, the sums that BatchNorm's
    backward needs (du, du * xhat per channel) are in ``state``'s workspace afterwards and ``slots`` says how many partial slots —
    hand it to ``bn_train_bwd(partial_slabs=)``; 0: nothing was written, the backward runs its own reduction.  Only valid when the
    gradient that convolution computes is the WHOLE gradient of that layer's output (no other consumer)."""

    def __init__(self, z: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, state: "BnState", act: bool):
        self.z, self.gamma, self.beta, self.state, self.act = z, gamma, beta, state, act
        self.slots = 0


def conv_stats_written() -> int:
    """Partial-sum slots this thread's last ``dy_conv2d_nhwc`` left in ``bn_stats`` (0: the launched kernel has no statistics epilogue)."""
    return int(lib().dy_conv_stats_written())


def conv2d(x: torch.Tensor, pc: PackedConv, out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           out_f32: bool = False, up2x: bool = False, x2: Optional[torch.Tensor] = None, dil2: bool = False,
           bn_stats: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, bn_behind: Optional[BnBehind] = None) -> torch.Tensor:
    """act(conv(x) + bias) (+ residual) through ``dy_conv2d_nhwc``.

    ``x2``: optional second input whose channels follow x's (Concat folded into the gather);
    ``up2x``: x is consumed through a fused 2x nearest upsample.
    ``dil2``: x is consumed zero-dilated by 2 (value at even (h, w) only): the gather of a stride-2 transposed conv.
    ``bn_stats``: the ``BnState`` of the train-mode BatchNorm that follows: a kernel with the statistics epilogue leaves the output's
    per-channel partial sums in its workspace (``conv_stats_written()`` = how many slots; hand that to ``bn_train_fwd(partial_slabs=)``).
    ``out_dtype``: storage type of the output when it differs from the input's (mixed-precision plans of BASELINE config 5: fp8 ->
    float16 behind an fp8 trunk, float16 -> fp8 into one; ``dy_conv_desc.y_dtype1``, the flat-K kernel).
    """
    require_device(x, "conv2d input")
    if x.dtype != pc.dtype:
        raise TypeError(f"conv2d: input dtype {x.dtype} != packed weight dtype {pc.dtype}")
    n, c1, hb, wb = x.shape
    h, w = (2 * hb, 2 * wb) if (up2x or dil2) else (hb, wb)
    cin = c1 + (x2.shape[1] if x2 is not None else 0)
    if cin != pc.cin:
        raise ValueError(f"conv2d: input has {cin} channels, weights expect {pc.cin}")
    ho, wo = conv_out_hw(h, w, pc.k, pc.stride, pc.pad)
    if out_dtype is not None and out_f32:
        raise ValueError("conv2d: out_dtype and out_f32 exclude each other")
    odt = torch.float32 if out_f32 else (out_dtype or x.dtype)
    if out is None:
        epc_o = chan_gran(odt)  # keep every pixel row 16-byte aligned (vector stores in all kernels)
        out = alloc_nhwc(n, pc.cout, ho, wo, odt, x.device, ld=-(-pc.cout // epc_o) * epc_o)
    elif tuple(out.shape) != (n, pc.cout, ho, wo) or out.dtype != odt:
        raise ValueError(f"conv2d: out has shape {tuple(out.shape)}/{out.dtype}, expected {(n, pc.cout, ho, wo)}/{odt}")
    xp, ldx = view_params(x)
    yp, ldy = view_params(out)
    if odt != x.dtype and not out_f32:
        pc = pc.rows()  # another storage type on the output is built in the flat-K kernel only (DY_WLAYOUT_ROWS)
    pc = pc.for_call(n * hb * wb * ldx * x.element_size(), n * ho * wo * ldy * out.element_size(), ldy, yp, residual is not None, out_f32,
                     x2 is not None or up2x or dil2)
    d = ConvDesc()
    d.x, d.w, d.bias, d.y = xp, pc.w.data_ptr(), pc.b.data_ptr(), yp
    d.batch, d.h, d.w_in, d.cin, d.ld_x = n, h, w, cin, ldx
    d.ho, d.wo, d.cout, d.ld_y = ho, wo, pc.cout, ldy
    d.ksize, d.stride, d.pad, d.groups = pc.k, pc.stride, pc.pad, pc.groups
    d.act, d.dtype, d.out_f32 = pc.act, dy_dtype(x.dtype), int(out_f32)
    d.k_pad, d.cout_pad, d.up2x, d.w_layout = pc.k_pad, pc.cout_pad, (2 if dil2 else int(up2x)), pc.layout
    if bn_stats is not None:
        if bn_stats.c != pc.cout:
            raise ValueError("conv2d: bn_stats must be the BnState of a BatchNorm over the output channels")
        d.bn_stats = bn_stats.ws.data_ptr()
    if bn_behind is not None:
        bb = bn_behind
        if bn_stats is not None or residual is not None or out_f32 or tuple(bb.z.shape) != tuple(out.shape) or bb.z.dtype != x.dtype or bb.state.c != pc.cout:
            raise ValueError("conv2d: bn_behind needs the saved z of the layer in front (shape / dtype of the output) and excludes bn_stats / residual / out_f32")
        if bb.gamma.dtype != torch.float32 or bb.beta.dtype != torch.float32 or not bb.gamma.is_contiguous() or not bb.beta.is_contiguous():
            raise ValueError("conv2d: bn_behind.gamma / beta must be contiguous fp32")
        d.bn_stats = bb.state.ws.data_ptr()
        d.bnb_z, d.bnb_ld_z = view_params(bb.z)
        d.bnb_act = DY_ACT_SILU if bb.act else DY_ACT_NONE
        d.bnb_mean, d.bnb_rstd, d.bnb_gamma, d.bnb_beta = bb.state.mean.data_ptr(), bb.state.rstd.data_ptr(), bb.gamma.data_ptr(), bb.beta.data_ptr()
    if residual is not None:
        if tuple(residual.shape) != tuple(out.shape) or residual.dtype != x.dtype:
            raise ValueError("conv2d: residual must match the output shape and the input dtype")
        d.residual, d.ld_res = view_params(residual)
    if x2 is not None:
        if tuple(x2.shape[2:]) != (h, w) or x2.shape[0] != n or x2.dtype != x.dtype:
            raise ValueError("conv2d: x2 must have the (upsampled) spatial size and dtype of x")
        d.x2, d.ld_x2 = view_params(x2)
        d.cin_split = c1
    if pc.wscale is not None:
        d.w_scale, d.act_scale = pc.wscale.data_ptr(), pc.act_scale
    if odt != x.dtype and not out_f32:
        d.y_dtype1 = dy_dtype(odt) + 1
        if odt == FP8:
            d.act_scale = fp8_act_scale()  # the output quantum (the input is 16-bit: no pack-time scale)
    _launch(lib().dy_conv2d_nhwc, (C.byref(d),), keep=(d, x, out, residual, x2, pc, bn_behind))
    if bn_behind is not None:
        bn_behind.slots = conv_stats_written()
    if _absmax_log is not None and not out_f32:  # activations that WOULD be stored in fp8 (the fp32 head logits are not)
        _absmax_log.append(out.float().abs().amax())
    return out


def quantize_fp8(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """16-bit / fp32 NHWC view -> fp8 (e4m3fn) with the network's activation scale, through ``dy_quantize_fp8_nhwc``."""
    require_device(x, "quantize input")
    n, c, h, w = x.shape
    if c % 16:
        raise ValueError("quantize_fp8: channels must be a multiple of 16 (one fp8 chunk)")
    if out is None:
        out = alloc_nhwc(n, c, h, w, FP8, x.device)
    (xp, ldx), (op, ldo) = view_params(x), view_params(out)
    _launch(lib().dy_quantize_fp8_nhwc, (xp, op, n * h * w, c, ldx, ldo, dy_dtype(x.dtype), fp8_act_scale()), keep=(x, out))
    return out


# ---- fused stem ---------------------------------------------------------------------------------------


class PackedStem:
    """Weights of the first layer Conv(cin<=3, cout, 3, 2) for ``dy_stem_conv3x3s2_nchw``: rows of the folded
    OIHW taps (k = c*9 + r*3 + q) zero padded to 32, rows padded to a multiple of 16 (include/dyolo.h)."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, act: bool, dtype: torch.dtype, device):
        cout, cin, k, k2 = weight.shape
        if not (k == k2 == 3 and cin * 9 <= 32 and cout <= 80):
            raise ValueError("PackedStem: needs a 3x3 kernel, cin <= 3 and cout <= 80")
        self.cout, self.cin, self.dtype = cout, cin, dtype
        self.act = _act_code(act)
        cp = -(-cout // 16) * 16
        wp = torch.zeros((cp, 32), dtype=torch.float32)
        wp[:cout, : cin * 9] = weight.detach().to(torch.float32).cpu().reshape(cout, cin * 9)
        bp = torch.zeros((cp,), dtype=torch.float32)
        bp[:cout] = bias.detach().to(torch.float32).cpu()
        if dtype == F16X2:  # [hi rows | lo rows | inverse row scales] in one buffer (include/dyolo.h); the split rules of PackedConv
            if cout % 8 or cout > 64:
                raise ValueError("PackedStem: split-float16 storage needs cout a multiple of 8 up to 64")
            sc = torch.exp2(13.0 - torch.floor(torch.log2(wp.abs().amax(1).clamp_min(1e-30))))
            ws = wp * sc[:, None]
            hi = torch.where(ws.abs() < 2.0 ** -14, torch.zeros_like(ws), ws).to(torch.float16)
            lo = (ws - hi.to(torch.float32)).to(torch.float16)
            inv = (1.0 / sc).to(torch.float32)
            raw = torch.cat((hi.reshape(-1).view(torch.uint8), lo.reshape(-1).view(torch.uint8), inv.view(torch.uint8)))
            self.w = raw.contiguous().to(device)
            self.b = bp.contiguous().to(device)
            return
        self.w = wp.to(dtype).contiguous().to(device)
        self.b = bp.contiguous().to(device)


def stem_conv(src: torch.Tensor, ps: PackedStem, out: Optional[torch.Tensor] = None, mark_input: bool = False) -> torch.Tensor:
    """fp32 NCHW image -> act(conv3x3 s2 p1 + bias) as an NHWC view of ``ps.dtype`` (layout cast fused in)."""
    require_device(src, "input")
    if src.dtype != torch.float32 or not src.is_contiguous() or src.dim() != 4 or src.shape[1] != ps.cin:
        raise ValueError(f"stem_conv expects a contiguous fp32 (N,{ps.cin},H,W) tensor")
    n, c, h, w = src.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if out is None:
        e = chan_gran(ps.dtype)
        out = alloc_nhwc(n, ps.cout, ho, wo, ps.dtype, src.device, ld=-(-ps.cout // e) * e)
    op, ld = view_params(out)
    args = (src.data_ptr(), ps.w.data_ptr(), ps.b.data_ptr(), op, n, c, h, w, ps.cout, ld, ps.act, dy_dtype(ps.dtype))
    _launch(lib().dy_stem_conv3x3s2_nchw, args, keep=(out, ps))
    if mark_input and _recording is not None:
        _recording.input_slot = (len(_recording.ops) - 1, 0)
    return out



_STEM_TRAIN_W: dict = {}


def stem_conv_u8(img: torch.Tensor, weight: torch.Tensor, dtype: torch.dtype, divisor: float = 255.0) -> torch.Tensor:
    """Training stem: uint8 NCHW image -> conv3x3 s2 p1 (no bias, no activation) as an NHWC view of ``dtype`` through
    ``dy_stem_conv3x3s2_nchw_u8``.  ``weight``: the fp32 master weight (cout, cin <= 3, 3, 3) on the device — its OIHW rows ARE the
    kernel's k order, so packing is one casting copy into a persistent zero-padded [cout][32] buffer."""
    require_device(img, "training image")
    cout, cin = weight.shape[0], weight.shape[1]
    if img.dtype != torch.uint8 or not img.is_contiguous() or img.shape[1] != cin or tuple(weight.shape[2:]) != (3, 3) or cout % 16 or cout > 80:
        raise ValueError("stem_conv_u8: contiguous uint8 (N, cin, H, W) image, (cout % 16 == 0, cin, 3, 3) weights")
    key = (weight.data_ptr(), cout, cin, dtype, str(img.device))  # (an address may be handed to another model's stem of another width later)
    ent = _STEM_TRAIN_W.get(key)
    if ent is None:
        ent = _STEM_TRAIN_W[key] = (torch.zeros((cout, 32), dtype=dtype, device=img.device), zero_bias(cout, img.device))
    wp, bp = ent
    wp[:, : cin * 9].copy_(weight.detach().reshape(cout, cin * 9))
    n, _, h, w = img.shape
    out = alloc_nhwc(n, cout, (h - 1) // 2 + 1, (w - 1) // 2 + 1, dtype, img.device)
    op, ld = view_params(out)
    _launch(lib().dy_stem_conv3x3s2_nchw_u8, (img.data_ptr(), float(divisor), wp.data_ptr(), bp.data_ptr(), op, n, cin, h, w, cout, ld, DY_ACT_NONE, dy_dtype(dtype)),
            keep=(img, wp, bp, out))
    return out


class PackedStem2:
    """Weights of layers 0 + 1 for ``dy_stem2_fused``: the stem rows as in :class:`PackedStem` and the stride-2 3x3
    layer as [64][288] rows with k = (r*3 + q)*32 + c (include/dyolo.h)."""

    def __init__(self, w0, b0, act0: bool, w1, b1, act1: bool, dtype: torch.dtype, device):
        if tuple(w0.shape) != (32, 3, 3, 3) or tuple(w1.shape) != (64, 32, 3, 3):
            raise ValueError("PackedStem2: built for Conv(3, 32, 3, 2) followed by a 3x3 stride-2 32 -> 64 layer")
        self.stem = PackedStem(w0, b0, act0, dtype, device)
        self.dtype = dtype
        self.act1 = _act_code(act1)
        self.w1_scale = None
        if dtype == F16X2:  # layer 1 as dy_conv2d_nhwc's split rows (K order (r, q, c), 8-channel groups [hi x 8 | lo x 8]) + inverse row scales
            self._pc1 = PackedConv(w1.detach().cpu(), b1.detach().cpu(), 2, 1, 1, act1, dtype, device)
            if self._pc1.cout_pad != 64 or self._pc1.k_pad != 288:
                raise ValueError("PackedStem2: unexpected split pack geometry")
            self.w1, self.b1, self.w1_scale = self._pc1.w, self._pc1.b, self._pc1.wscale
            return
        self.w1 = w1.detach().to(torch.float32).cpu().permute(0, 2, 3, 1).reshape(64, 288).to(dtype).contiguous().to(device)
        self.b1 = b1.detach().to(torch.float32).cpu().contiguous().to(device)


def stem2_fused_supported(cin: int, c0: int, c1: int, h: int, w: int, dtype: torch.dtype) -> bool:
    return dtype != FP8 and bool(lib().dy_stem2_fused_supported(cin, c0, c1, h, w, dy_dtype(dtype)))


def stem2_fused(src: torch.Tensor, ps: PackedStem2, out: Optional[torch.Tensor] = None, mark_input: bool = False) -> torch.Tensor:
    """fp32 NCHW image -> layers 0 and 1 (two stride-2 3x3 convolutions with SiLU) as an NHWC view at 1/4 resolution."""
    require_device(src, "input")
    if src.dtype != torch.float32 or not src.is_contiguous() or src.dim() != 4 or src.shape[1] != 3:
        raise ValueError("stem2_fused expects a contiguous fp32 (N,3,H,W) tensor")
    n, _, h, w = src.shape
    if out is None:
        out = alloc_nhwc(n, 64, h // 4, w // 4, ps.dtype, src.device)
    op, ld = view_params(out)
    d = Stem2Desc(x=src.data_ptr(), w0=ps.stem.w.data_ptr(), b0=ps.stem.b.data_ptr(), w1=ps.w1.data_ptr(), b1=ps.b1.data_ptr(), y=op,
                  n=n, h=h, w=w, ld_y=ld, act0=ps.stem.act, act1=ps.act1, dtype=dy_dtype(ps.dtype),
                  w1_scale=ps.w1_scale.data_ptr() if ps.w1_scale is not None else None)
    _launch(lib().dy_stem2_fused, (C.byref(d),), keep=(d, out, ps))
    if mark_input and _recording is not None:
        _recording.input_slot = (len(_recording.ops) - 1, -1)
    return out

# ---- layout ops -------------------------------------------------------------------------------------


def to_nhwc(src: torch.Tensor, dtype: torch.dtype, c_pad: Optional[int] = None, out: Optional[torch.Tensor] = None,
            mark_input: bool = False) -> torch.Tensor:
    """fp32 NCHW (contiguous) -> NHWC view of ``dtype`` with channels zero-padded to ``c_pad``."""
    require_device(src, "input")
    if src.dtype != torch.float32 or not src.is_contiguous():
        raise ValueError("to_nhwc expects a contiguous fp32 NCHW tensor")
    n, c, h, w = src.shape
    epc = chan_gran(dtype)
    c_pad = c_pad or (c + epc - 1) // epc * epc
    if out is None:
        out = alloc_nhwc(n, c_pad, h, w, dtype, src.device)
    op, ld = view_params(out)
    args = (src.data_ptr(), op, n, c, h, w, c_pad, ld, dy_dtype(dtype))
    _launch(lib().dy_nchw_f32_to_nhwc, args, keep=(out,))
    if mark_input and _recording is not None:
        _recording.input_slot = (len(_recording.ops) - 1, 0)
    return out


def u8_to_nhwc(src: torch.Tensor, dtype: torch.dtype, divisor: float = 255.0) -> torch.Tensor:
    """uint8 NCHW (contiguous) -> NHWC view of ``dtype`` holding src / divisor, channels zero-padded to one chunk."""
    require_device(src, "input")
    if src.dtype != torch.uint8 or not src.is_contiguous() or src.dim() != 4:
        raise ValueError("u8_to_nhwc expects a contiguous uint8 NCHW tensor")
    n, c, h, w = src.shape
    epc = elems_per_chunk(dtype)
    c_pad = (c + epc - 1) // epc * epc
    out = alloc_nhwc(n, c_pad, h, w, dtype, src.device)
    op, ld = view_params(out)
    _launch(lib().dy_nchw_u8_to_nhwc, (src.data_ptr(), op, n, c, h, w, c_pad, ld, divisor, dy_dtype(dtype)), keep=(src, out))
    return out


def to_nchw_f32(x: torch.Tensor) -> torch.Tensor:
    """NHWC view -> contiguous fp32 NCHW tensor (module-level parity checks, user hand-back)."""
    require_device(x)
    n, c, h, w = x.shape
    xp, ld = view_params(x)
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    _launch(lib().dy_nhwc_to_nchw_f32, (xp, out.data_ptr(), n, c, h, w, ld, dy_dtype(x.dtype)), keep=(x, out))
    return out


def upsample2x(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    require_device(x)
    n, c, h, w = x.shape
    if out is None:
        out = alloc_nhwc(n, c, 2 * h, 2 * w, x.dtype, x.device)
    xp, lds = view_params(x)
    op, ldd = view_params(out)
    _launch(lib().dy_upsample2x_nhwc, (xp, op, n, h, w, c, lds, ldd, dy_dtype(x.dtype)), keep=(x, out))
    return out


def copy_nhwc(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    require_device(x)
    n, c, h, w = x.shape
    xp, lds = view_params(x)
    op, ldd = view_params(out)
    _launch(lib().dy_copy_nhwc, (xp, op, n, h, w, c, lds, ldd, dy_dtype(x.dtype)), keep=(x, out))
    return out


def sppf_maxpool3(x: torch.Tensor, y1: torch.Tensor, y2: torch.Tensor, y3: torch.Tensor, k: int) -> None:
    require_device(x)
    n, c, h, w = x.shape
    xp, ld = view_params(x)
    ptrs = []
    for y in (y1, y2, y3):
        p, l2 = view_params(y)
        if l2 != ld or tuple(y.shape) != tuple(x.shape):
            raise ValueError("sppf_maxpool3: outputs must be slices of the same buffer as x")
        ptrs.append(p)
    _launch(lib().dy_sppf_maxpool3, (xp, *ptrs, n, h, w, c, ld, k, dy_dtype(x.dtype)), keep=(x, y1, y2, y3))


# ---- detect decode + NMS --------------------------------------------------------------------------


def detect_decode(levels: Sequence[torch.Tensor], strides: Sequence[float], nc: int, reg_max: int,
                  out: Optional[torch.Tensor] = None, nms_bufs: Optional["NmsBuffers"] = None, conf_thres: float = 0.25,
                  classes_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """levels[i]: fp32 NHWC view (N, 4*reg_max+nc, H_i, W_i) -> (N, 4+nc, A) fp32.

    ``nms_bufs``: fuse the NMS candidate filter into this pass (then call ``nms(..., prefiltered=True)``
    with the same buffers, conf_thres and classes_mask)."""
    n = levels[0].shape[0]
    d = DecodeDesc()
    A = 0
    for i, t in enumerate(levels):
        require_device(t)
        if t.dtype != torch.float32 or t.shape[1] != 4 * reg_max + nc:
            raise ValueError("detect_decode: levels must be fp32 with 4*reg_max+nc channels")
        p, ld = view_params(t)
        d.level[i], d.h[i], d.w[i], d.ld[i], d.stride[i] = p, t.shape[2], t.shape[3], ld, float(strides[i])
        A += t.shape[2] * t.shape[3]
    d.n_levels, d.batch, d.nc, d.reg_max = len(levels), n, nc, reg_max
    if out is None:
        out = torch.empty((n, 4 + nc, A), dtype=torch.float32, device=levels[0].device)
    d.out = out.data_ptr()
    if nms_bufs is not None:
        if (nms_bufs.batch, nms_bufs.anchors) != (n, A):
            raise ValueError("detect_decode: nms_bufs were sized for another batch / anchor count")
        d.nms_workspace, d.nms_workspace_bytes = nms_bufs.workspace.data_ptr(), nms_bufs.workspace.numel()
        d.conf_thres = conf_thres
        d.classes_mask = classes_mask.data_ptr() if classes_mask is not None else None
    _launch(lib().dy_detect_decode, (C.byref(d),), keep=(d, out, nms_bufs, classes_mask, *levels))
    return out


def branch_fused_supported(c_in: int, c_mid: int, c_out: int, kind: int, nc: int, reg_max: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and bool(lib().dy_detect_branch_fused_supported(c_in, c_mid, c_out, kind, nc, reg_max, dy_dtype(dtype)))


def nms_reset_counts(bufs: "NmsBuffers") -> None:
    """Zero the per-image candidate counts: once per pass, before the first class branch appends candidates."""
    _launch(lib().dy_nms_reset_counts, (bufs.workspace.data_ptr(), bufs.batch), keep=(bufs,))


def detect_branch_fused(x: torch.Tensor, pc3: "PackedConv", w1: torch.Tensor, b1: torch.Tensor, kind: int, nc: int, reg_max: int, stride: float, pred: torch.Tensor,
                        anchor0: int, nms_bufs: Optional["NmsBuffers"] = None, conf_thres: float = 0.25, classes_mask: Optional[torch.Tensor] = None) -> None:
    """One Detect branch (kind 1 = box, 2 = class) of one level from its second 3x3 conv to rows of ``pred`` through
    ``dy_detect_branch_fused``.  x: the branch's first conv output (N, 64, H, W); pc3: the 3x3 Conv packed in
    DY_WLAYOUT_HALO3X3; (w1, b1): ``pack_frag1x1`` of the plain 1x1 conv."""
    require_device(x, "branch input")
    if pc3.layout != _lib.DY_WLAYOUT_HALO3X3 or pc3.dtype != x.dtype or pc3.k != 3 or pc3.stride != 1:
        raise ValueError("detect_branch_fused: the 3x3 conv must be packed in DY_WLAYOUT_HALO3X3 for x's dtype")
    n, c, h, w = x.shape
    d = BranchDesc()
    d.x, d.ld_x = view_params(x)
    d.w3, d.b3, d.w1, d.b1, d.out = pc3.w.data_ptr(), pc3.b.data_ptr(), w1.data_ptr(), b1.data_ptr(), pred.data_ptr()
    d.batch, d.h, d.w, d.c_in, d.c_mid, d.nc, d.reg_max, d.kind, d.dtype = n, h, w, c, pc3.cout, nc, reg_max, kind, dy_dtype(x.dtype)
    d.anchors, d.anchor0, d.stride = pred.shape[2], anchor0, float(stride)
    if pc3.act not in (DY_ACT_SILU, DY_ACT_SILU_L2E):
        raise ValueError("detect_branch_fused: the 3x3 conv must end in SiLU")
    d.act_l2e = int(pc3.act == DY_ACT_SILU_L2E)  # (the caller packed w1 divided by log2 e then: the logits stay in true units)
    if pred.dtype != torch.float32 or pred.shape[0] != n or pred.shape[1] != 4 + nc or not pred.is_contiguous():
        raise ValueError("detect_branch_fused: pred must be a contiguous fp32 (N, 4 + nc, A) tensor")
    if kind == 2 and nms_bufs is not None:
        if (nms_bufs.batch, nms_bufs.anchors) != (n, pred.shape[2]):
            raise ValueError("detect_branch_fused: nms_bufs were sized for another batch / anchor count")
        d.nms_workspace, d.nms_workspace_bytes = nms_bufs.workspace.data_ptr(), nms_bufs.workspace.numel()
        d.conf_thres = conf_thres
        d.classes_mask = classes_mask.data_ptr() if classes_mask is not None else None
    _launch(lib().dy_detect_branch_fused, (C.byref(d),), keep=(d, x, pc3, w1, b1, pred, nms_bufs, classes_mask))


def pack_frag1x1(weight: torch.Tensor, bias: torch.Tensor, dtype: torch.dtype, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """(cout, cin[,1,1]) weights + bias in DY_WLAYOUT_FRAG1X1 order (include/dyolo.h), whatever kernel PackedConv would pick."""
    if dtype == F16X2:
        # split float16 (dy_detect_head_decode with DY_F16X2): the float16 fragment image of the hi halves, then of the lo halves (rows scaled by
        # a power of two into [2^13, 2^14) first, hi flushed to 0 below float16's smallest normal: PackedConv's rule); bias followed by the
        # inverse row scales
        w2 = weight.detach().to(torch.float32).cpu().reshape(weight.shape[0], -1)
        sc = torch.exp2(13.0 - torch.floor(torch.log2(w2.abs().amax(1).clamp_min(1e-30))))
        ws = w2 * sc[:, None]
        hi = torch.where(ws.abs() < 2.0 ** -14, torch.zeros_like(ws), ws).to(torch.float16)
        lo = (ws - hi.to(torch.float32)).to(torch.float16)
        zb = torch.zeros(weight.shape[0])
        wh, bp = pack_frag1x1(hi.to(torch.float32), bias, torch.float16, "cpu")
        wl, _ = pack_frag1x1(lo.to(torch.float32), zb, torch.float16, "cpu")
        sp = torch.ones_like(bp)
        sp[: weight.shape[0]] = 1.0 / sc
        return torch.cat((wh, wl)).contiguous().to(device), torch.cat((bp, sp)).contiguous().to(device)
    e = elems_per_chunk(dtype)
    w2 = weight.detach().to(torch.float32).cpu().reshape(weight.shape[0], -1)
    cout, cin = w2.shape
    kc, bn = 4 * e, (128 if cout > 64 else (64 if cout > 16 else 16))
    nt, nkg = -(-cout // bn), -(-cin // kc)
    wpad = torch.zeros((nt * bn, nkg * kc), dtype=torch.float32)
    wpad[:cout, :cin] = w2
    wp = wpad.view(nt, bn // 16, 16, nkg, 4, e).permute(0, 3, 1, 4, 2, 5).contiguous().view(-1)
    bp = torch.zeros((nt * bn,), dtype=torch.float32)
    bp[:cout] = bias.detach().to(torch.float32).cpu()
    return wp.to(dtype).contiguous().to(device), bp.to(device)


def head_decode_supported(c_box: int, c_cls: int, nc: int, reg_max: int, dtype: torch.dtype) -> bool:
    return dtype != FP8 and bool(lib().dy_detect_head_decode_supported(c_box, c_cls, nc, reg_max, dy_dtype(dtype)))


def detect_head_decode(x_box: Sequence[torch.Tensor], x_cls: Sequence[torch.Tensor], packed_box, packed_cls,
                       strides: Sequence[float], nc: int, reg_max: int, out: Optional[torch.Tensor] = None,
                       nms_bufs: Optional["NmsBuffers"] = None, conf_thres: float = 0.25,
                       classes_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Last 1x1 convs of both Detect branches + decode (+ NMS candidate filter) in one ``dy_detect_head_decode`` pass.

    x_box[i] / x_cls[i]: NHWC views (N, c, H_i, W_i), the inputs of cv2[i][2] / cv3[i][2]; packed_box[i] / packed_cls[i]:
    ``pack_frag1x1`` (w, b) of those convs.  Returns (N, 4+nc, A) fp32."""
    n = x_box[0].shape[0]
    d = HeadDecodeDesc()
    A = 0
    for i, (tb, tc) in enumerate(zip(x_box, x_cls)):
        require_device(tb)
        require_device(tc)
        if tb.dtype != tc.dtype or tb.shape[2:] != tc.shape[2:] or tb.shape[0] != n or tc.shape[0] != n:
            raise ValueError("detect_head_decode: branch inputs of one level must agree in dtype and size")
        if tb.shape[1] != x_box[0].shape[1] or tc.shape[1] != x_cls[0].shape[1] or tb.dtype != x_box[0].dtype:
            raise ValueError("detect_head_decode: all levels must have the same branch widths and dtype")
        (d.x_box[i], d.ld_box[i]), (d.x_cls[i], d.ld_cls[i]) = view_params(tb), view_params(tc)
        d.w_box[i], d.b_box[i] = packed_box[i][0].data_ptr(), packed_box[i][1].data_ptr()
        d.w_cls[i], d.b_cls[i] = packed_cls[i][0].data_ptr(), packed_cls[i][1].data_ptr()
        d.h[i], d.w[i], d.stride[i] = tb.shape[2], tb.shape[3], float(strides[i])
        A += tb.shape[2] * tb.shape[3]
    d.n_levels, d.batch, d.nc, d.reg_max = len(x_box), n, nc, reg_max
    d.c_box, d.c_cls, d.dtype = x_box[0].shape[1], x_cls[0].shape[1], dy_dtype(x_box[0].dtype)
    if out is None:
        out = torch.empty((n, 4 + nc, A), dtype=torch.float32, device=x_box[0].device)
    d.out = out.data_ptr()
    if nms_bufs is not None:
        if (nms_bufs.batch, nms_bufs.anchors) != (n, A):
            raise ValueError("detect_head_decode: nms_bufs were sized for another batch / anchor count")
        d.nms_workspace, d.nms_workspace_bytes = nms_bufs.workspace.data_ptr(), nms_bufs.workspace.numel()
        d.conf_thres = conf_thres
        d.classes_mask = classes_mask.data_ptr() if classes_mask is not None else None
    _launch(lib().dy_detect_head_decode, (C.byref(d),),
            keep=(d, out, nms_bufs, classes_mask, *x_box, *x_cls, *packed_box, *packed_cls))
    return out


class NmsBuffers:
    """Persistent outputs + workspace of ``dy_nms`` for one (batch, anchors, max_det)."""

    def __init__(self, batch: int, anchors: int, max_det: int, device, candidates_per_anchor: int = 1):
        """``candidates_per_anchor``: nc for the validator's multi_label NMS (one candidate per (anchor, class) pair), else 1."""
        self.batch, self.anchors, self.max_det, self.cpa = batch, anchors, max_det, candidates_per_anchor
        nbytes = lib().dy_nms_workspace_bytes(batch, anchors * candidates_per_anchor)
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        self.out = torch.empty((batch, max_det, 6), dtype=torch.float32, device=device)
        self.count = torch.empty((batch,), dtype=torch.int32, device=device)
        self.index = torch.empty((batch, max_det), dtype=torch.int32, device=device)


def nms(pred: torch.Tensor, conf_thres: float, iou_thres: float, max_det: int = 300, max_nms: int = 30000,
        max_wh: float = 7680.0, agnostic: bool = False, nc: int = 0, classes_mask: Optional[torch.Tensor] = None,
        bufs: Optional[NmsBuffers] = None, prefiltered: bool = False, multi_label: bool = False) -> NmsBuffers:
    """Batched NMS on (N, 4+nc(+nm), A) fp32 predictions; results stay on the device.
    ``prefiltered``: ``bufs.workspace`` already holds the candidates (detect_decode(nms_bufs=bufs)).
    ``multi_label``: the validator's form (utils/ops.py:286-288): every (anchor, class) pair above conf is a candidate."""
    require_device(pred, "prediction")
    if pred.dtype != torch.float32 or not pred.is_contiguous() or pred.dim() != 3:
        raise ValueError("nms expects a contiguous fp32 (N, 4+nc, A) tensor")
    n, ch, A = pred.shape
    nc = nc or ch - 4
    cpa = nc if (multi_label and nc > 1) else 1
    if bufs is None or (bufs.batch, bufs.anchors, bufs.max_det) != (n, A, max_det) or getattr(bufs, "cpa", 1) < cpa:
        if prefiltered:
            raise ValueError("nms(prefiltered=True) needs the NmsBuffers that detect_decode filled")
        bufs = NmsBuffers(n, A, max_det, pred.device, candidates_per_anchor=cpa)
    d = NmsDesc()
    d.pred, d.batch, d.nc, d.n_extra, d.anchors = pred.data_ptr(), n, nc, ch - 4 - nc, A
    d.conf_thres, d.iou_thres, d.max_det, d.max_nms = conf_thres, iou_thres, max_det, max_nms
    d.max_wh, d.agnostic = max_wh, int(agnostic)
    d.classes_mask = classes_mask.data_ptr() if classes_mask is not None else None
    d.out, d.out_count, d.out_index = bufs.out.data_ptr(), bufs.count.data_ptr(), bufs.index.data_ptr()
    d.workspace, d.workspace_bytes = bufs.workspace.data_ptr(), bufs.workspace.numel()
    d.prefiltered, d.multi_label = int(prefiltered), int(bool(multi_label))
    _launch(lib().dy_nms, (C.byref(d),), keep=(d, pred, bufs, classes_mask))
    return bufs


def scale_boxes_(bufs: NmsBuffers, params: torch.Tensor) -> None:
    """In-place scale_boxes + clip_boxes of the kept rows; params: device fp32 (N,5)."""
    _launch(lib().dy_scale_boxes, (bufs.out.data_ptr(), bufs.count.data_ptr(), params.data_ptr(), bufs.batch, bufs.max_det),
            keep=(bufs, params))


# ---- training loss (forward) ----------------------------------------------------------------------------


def detection_loss(levels: Sequence[torch.Tensor], gt: torch.Tensor, strides: Sequence[float], nc: int, reg_max: int = 16,
                   topk: int = 10, alpha: float = 0.5, beta: float = 6.0, box: float = 7.5, cls: float = 0.5, dfl: float = 1.5,
                   want_owner: bool = False, want_grad: bool = False):
    """v8DetectionLoss forward through ``dy_detection_loss``.

    levels[i]: fp32 NHWC view (N, 4*reg_max+nc, H_i, W_i) — Detect's training-mode outputs; gt: device fp32 (N, gmax, 5)
    [cls, x1, y1, x2, y2] in pixels, zero rows = padding.  Returns (out[4] = box, cls, dfl, total, owner or None) and,
    with ``want_grad``, a third item: the list of d total / d levels[i] (fp32 NHWC views shaped like the inputs)."""
    n = levels[0].shape[0]
    d = LossDesc()
    A = 0
    for i, t in enumerate(levels):
        require_device(t)
        if t.dtype != torch.float32 or t.shape[1] != 4 * reg_max + nc:
            raise ValueError("detection_loss: levels must be fp32 with 4*reg_max+nc channels")
        p_, ld = view_params(t)
        d.level[i], d.h[i], d.w[i], d.ld[i], d.stride[i] = p_, t.shape[2], t.shape[3], ld, float(strides[i])
        A += t.shape[2] * t.shape[3]
    dev = levels[0].device
    gt = gt.to(dev, torch.float32).contiguous()
    if gt.dim() != 3 or gt.shape[0] != n or gt.shape[2] != 5:
        raise ValueError("detection_loss: gt must be (N, gmax, 5)")
    gmax = gt.shape[1]
    d.n_levels, d.batch, d.nc, d.reg_max = len(levels), n, nc, reg_max
    d.gt, d.gmax, d.topk = (gt.data_ptr() if gmax else None), gmax, topk
    d.alpha, d.beta, d.box_gain, d.cls_gain, d.dfl_gain = alpha, beta, box, cls, dfl
    out = torch.empty(4, dtype=torch.float32, device=dev)
    owner = torch.empty((n, A), dtype=torch.int32, device=dev) if want_owner else None
    ws = torch.empty(lib().dy_detection_loss_workspace_bytes(n, A, gmax, topk), dtype=torch.uint8, device=dev)
    d.out, d.out_owner = out.data_ptr(), (owner.data_ptr() if want_owner else None)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    grads = None
    if want_grad:
        grads = []
        for i, t in enumerate(levels):
            gt_ = alloc_nhwc(n, t.shape[1], t.shape[2], t.shape[3], torch.float32, dev, ld=d.ld[i])
            d.grad_level[i], d.ld_grad[i] = view_params(gt_)
            grads.append(gt_)
    _launch(lib().dy_detection_loss, (C.byref(d),), keep=(d, out, owner, ws, gt, *levels, *(grads or ())))
    return (out, owner, grads) if want_grad else (out, owner)


# ---- train-mode BatchNorm (+ SiLU) -------------------------------------------------------------------------------


class BnState:
    """Saved batch statistics + workspace of one BatchNorm layer for one training step."""

    def __init__(self, c: int, device):
        self.c = c
        self.mean = torch.empty(c, dtype=torch.float32, device=device)
        self.rstd = torch.empty(c, dtype=torch.float32, device=device)
        self.ws = torch.empty(lib().dy_bn_workspace_bytes(c), dtype=torch.uint8, device=device)


def _rows(t: torch.Tensor) -> int:
    return t.shape[0] * t.shape[2] * t.shape[3]


def bn_train_fwd(z: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, state: BnState, act: bool, eps: float = 1e-3,
                 momentum: float = 0.03, running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None,
                 addend: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, partial_slabs: int = 0) -> torch.Tensor:
    """y = act(BN_batchstats(z) (+ addend)) through ``dy_bn_train_fwd``; z, y, addend: NHWC views (N, C, H, W).
    ``partial_slabs``: > 0 = ``state``'s workspace already holds that many partial-sum slots of z (``conv2d(bn_stats=state)``): no reduction pass."""
    require_device(z, "bn input")
    n, c, h, w = z.shape
    if out is None:
        out = alloc_nhwc(n, c, h, w, z.dtype, z.device)
    d = BnDesc()
    (d.z, d.ld_z), (d.y, d.ld_y) = view_params(z), view_params(out)
    if addend is not None:
        d.addend, d.ld_add = view_params(addend)
    d.rows, d.c, d.dtype, d.act = _rows(z), c, dy_dtype(z.dtype), DY_ACT_SILU if act else DY_ACT_NONE
    d.gamma, d.beta, d.mean, d.rstd = gamma.data_ptr(), beta.data_ptr(), state.mean.data_ptr(), state.rstd.data_ptr()
    if running_mean is not None:
        d.running_mean, d.running_var = running_mean.data_ptr(), running_var.data_ptr()
    d.eps, d.momentum = eps, momentum
    d.workspace, d.workspace_bytes = state.ws.data_ptr(), state.ws.numel()
    d.partial_slabs = int(partial_slabs)
    _launch(lib().dy_bn_train_fwd, (C.byref(d),), keep=(d, z, out, addend, gamma, beta, state, running_mean, running_var))
    return out


def bn_train_bwd(dy: torch.Tensor, z: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, state: BnState, act: bool,
                 dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 partial_slabs: int = 0):
    """(dz, dgamma, dbeta) of y = act(BN_batchstats(z)) given dy, through ``dy_bn_train_bwd`` (state from the forward).
    ``partial_slabs``: > 0 = the workspace already holds that many slots of the sums of du and du * xhat (``BnBehind.slots``): no reduction pass."""
    n, c, h, w = z.shape
    if out is None:
        out = alloc_nhwc(n, c, h, w, z.dtype, z.device)
    if dgamma is None:
        dgamma = torch.empty(c, dtype=torch.float32, device=z.device)
    if dbeta is None:
        dbeta = torch.empty(c, dtype=torch.float32, device=z.device)
    d = BnDesc()
    (d.z, d.ld_z), (d.dy, d.ld_dy), (d.dz, d.ld_dz) = view_params(z), view_params(dy), view_params(out)
    d.rows, d.c, d.dtype, d.act = _rows(z), c, dy_dtype(z.dtype), DY_ACT_SILU if act else DY_ACT_NONE
    d.gamma, d.beta, d.mean, d.rstd = gamma.data_ptr(), beta.data_ptr(), state.mean.data_ptr(), state.rstd.data_ptr()
    d.dgamma, d.dbeta = dgamma.data_ptr(), dbeta.data_ptr()
    d.workspace, d.workspace_bytes = state.ws.data_ptr(), state.ws.numel()
    d.partial_slabs = int(partial_slabs)
    _launch(lib().dy_bn_train_bwd, (C.byref(d),), keep=(d, z, dy, out, gamma, beta, state, dgamma, dbeta))
    return out, dgamma, dbeta


def silu_fwd(u: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, c, h, w = u.shape
    if out is None:
        out = alloc_nhwc(n, c, h, w, u.dtype, u.device)
    (up, ldu), (op, ldo) = view_params(u), view_params(out)
    _launch(lib().dy_silu_fwd, (up, op, _rows(u), c, ldu, ldo, dy_dtype(u.dtype)), keep=(u, out))
    return out


def silu_bwd(u: torch.Tensor, dy: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, c, h, w = u.shape
    if out is None:
        out = alloc_nhwc(n, c, h, w, u.dtype, u.device)
    (up, ldu), (dp, ldd), (op, ldo) = view_params(u), view_params(dy), view_params(out)
    _launch(lib().dy_silu_bwd, (up, dp, op, _rows(u), c, ldu, ldd, ldo, dy_dtype(u.dtype)), keep=(u, dy, out))
    return out


# ---- convolution gradients ----------------------------------------------------------------------------------------


# ---- weight gradients beside the input gradients (r03) -------------------------------------------------------------------
# In the backward of a layer the weight gradient (x, dz -> dw) feeds nothing but the parameter's gradient, while the input gradient
# (dz, w -> dx) is what the rest of the backward waits for.  The wgrad kernels are bound by their closing fp32 atomics and by latency
# (150 - 420 TFLOP/s), the dgrad / BatchNorm kernels by MFMA and HBM: issued on a SECOND stream (forked after dz exists, joined before
# the gradients are read) the two can overlap; inside the captured step graph a fork / join becomes parallel branches.
# MEASURED (r03, B = 64, graphed step): 37.5 ms with everything on one stream, 38.4 - 40.5 ms with the weight gradients on the side
# stream — the hipGraph's branches do not run beside each other to any advantage (the wgrad atomics and the BatchNorm passes contend
# for the same memory pipeline) — so it is OFF unless DYOLO_WGRAD_SIDE=1 asks for it.
_WGRAD_SIDE = {"on": os.environ.get("DYOLO_WGRAD_SIDE", "0") == "1", "streams": {}, "used": False}


def _side_stream(device) -> "torch.cuda.Stream":
    st = _WGRAD_SIDE["streams"].get(str(device))
    if st is None:
        st = _WGRAD_SIDE["streams"][str(device)] = torch.cuda.Stream(device=device)
    return st


def join_side_stream(device=None) -> None:
    """Make the current stream wait for every weight-gradient kernel issued on the side stream (call before the gradients / the sink are
    read: the sink flush, a bucket's flush, the optimizer)."""
    if not _WGRAD_SIDE["used"]:
        return
    for st in _WGRAD_SIDE["streams"].values():
        torch.cuda.current_stream(st.device).wait_stream(st)
    _WGRAD_SIDE["used"] = False


def conv_wgrad_into(x: torch.Tensor, dz: torch.Tensor, ksize: int, stride: int, pad: int, out: torch.Tensor) -> None:
    """``conv_wgrad(..., out=out)`` on the side stream (see above) when that is enabled; ``out`` must be a buffer nobody reads before
    ``join_side_stream`` (a trainer's gradient sink)."""
    if not _WGRAD_SIDE["on"]:
        conv_wgrad(x, dz, ksize, stride, pad, out=out)
        return
    side = _side_stream(x.device)
    side.wait_stream(torch.cuda.current_stream(x.device))  # x and dz exist
    with torch.cuda.stream(side):
        conv_wgrad(x, dz, ksize, stride, pad, out=out)
    x.record_stream(side)  # the allocator must not hand their memory out again before the side stream is past this point
    dz.record_stream(side)
    _WGRAD_SIDE["used"] = True


def conv_wgrad(x: torch.Tensor, dz: torch.Tensor, ksize: int, stride: int, pad: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d loss / d weight of z = conv2d(x, w, stride, pad): fp32 (cout, cin, k, k), through ``dy_conv2d_wgrad_nhwc``.
    ``out``: a contiguous fp32 (cout, k, k, cin) tensor the kernel ADDS into (a trainer's gradient sink) instead of a fresh zeroed one."""
    require_device(x, "wgrad input")
    n, cin, h, w = x.shape
    cout, ho, wo = dz.shape[1], dz.shape[2], dz.shape[3]
    if (ho, wo) != conv_out_hw(h, w, ksize, stride, pad) or dz.shape[0] != n or dz.dtype != x.dtype:
        raise ValueError("conv_wgrad: dz does not match the forward geometry / dtype")
    d = ConvDesc()
    (d.x, d.ld_x), (dzp, lddz) = view_params(x), view_params(dz)
    d.batch, d.h, d.w_in, d.cin, d.ho, d.wo, d.cout = n, h, w, cin, ho, wo, cout
    d.ksize, d.stride, d.pad, d.groups, d.dtype = ksize, stride, pad, 1, dy_dtype(x.dtype)
    if out is not None:
        if tuple(out.shape) != (cout, ksize, ksize, cin) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("conv_wgrad: out must be a contiguous fp32 (cout, k, k, cin) tensor")
        dw = out
    else:
        dw = torch.zeros((cout, ksize, ksize, cin), dtype=torch.float32, device=x.device)
    need = lib().dy_conv2d_wgrad_workspace_bytes(C.byref(d), lddz)
    if need < 0:
        check(int(need), "dy_conv2d_wgrad_workspace_bytes")
    ws = _wgrad_workspace(x.device, need) if need else None
    _launch(lib().dy_conv2d_wgrad_nhwc_ws, (C.byref(d), dzp, lddz, dw.data_ptr(), ws.data_ptr() if need else None, ws.numel() if need else 0),
            keep=(d, x, dz, dw, ws))
    return dw.permute(0, 3, 1, 2)


_WGRAD_WS: dict = {}
_WGRAD_WS_RETIRED: list = []


def _wgrad_workspace(device, need: int) -> torch.Tensor:
    """Scratch of ``dy_conv2d_wgrad_nhwc_ws`` (the 3x3 kernel's per-slab partial sums, <= ~40 MB at any shape): one buffer per
    (device, stream) -- calls that share it are ordered on that stream -- grown on demand and never shrunk."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _WGRAD_WS_RETIRED.append(ws)  # a captured step graph may still replay launches that write the smaller buffer: it stays allocated
        ws = _WGRAD_WS[key] = torch.empty(max(need, 48 << 20), dtype=torch.uint8, device=device)
    return ws


def conv_grouped_bwd(x: torch.Tensor, dz: torch.Tensor, weight: torch.Tensor, stride: int, pad: int, groups: int, need_dx: bool = True):
    """(dw, dx) of z = conv2d(x, weight, stride, pad, groups=groups) given dz through ``dy_conv2d_grouped_bwd_nhwc``.
    weight: the fp32 master tensor (cout, cin/groups, k, k) on the device; dw comes back in the same layout (fp32)."""
    require_device(x, "grouped wgrad input")
    n, cin, h, w = x.shape
    cout, ho, wo = dz.shape[1], dz.shape[2], dz.shape[3]
    k = weight.shape[2]
    if (ho, wo) != conv_out_hw(h, w, k, stride, pad) or dz.shape[0] != n or dz.dtype != x.dtype or weight.dtype != torch.float32 or not weight.is_contiguous():
        raise ValueError("conv_grouped_bwd: dz / weight do not match the forward geometry, dtype or layout")
    d = ConvDesc()
    (d.x, d.ld_x), (dzp, lddz) = view_params(x), view_params(dz)
    d.batch, d.h, d.w_in, d.cin, d.ho, d.wo, d.cout = n, h, w, cin, ho, wo, cout
    d.ksize, d.stride, d.pad, d.groups, d.dtype = k, stride, pad, groups, dy_dtype(x.dtype)
    dw = torch.zeros_like(weight)
    dx = alloc_nhwc(n, cin, h, w, x.dtype, x.device) if need_dx else None
    dxp, lddx = view_params(dx) if need_dx else (None, 0)
    _launch(lib().dy_conv2d_grouped_bwd_nhwc, (C.byref(d), dzp, lddz, weight.data_ptr(), dw.data_ptr(), dxp, lddx, None, 0), keep=(d, x, dz, weight, dw, dx))
    return dw, dx


def colsum(z: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-channel sum over (N, H, W) of an NHWC view: the bias gradient of a plain convolution.  ``out``: a contiguous fp32 (c,) tensor
    the sums are ADDED to (a trainer's gradient sink) instead of a fresh zeroed one."""
    zp, ld = view_params(z)
    if out is None:
        out = torch.zeros(z.shape[1], dtype=torch.float32, device=z.device)
    elif out.dtype != torch.float32 or out.numel() != z.shape[1] or not out.is_contiguous():
        raise ValueError("colsum: out must be a contiguous fp32 (c,) tensor")
    _launch(lib().dy_colsum, (zp, out.data_ptr(), _rows(z), z.shape[1], ld, dy_dtype(z.dtype)), keep=(z, out))
    return out


def pack_dgrad(weight: torch.Tensor, stride: int, dtype: torch.dtype, device, no_accumulate: bool = False) -> PackedConv:
    """Weights of the convolution that computes dx from dz: w'[ci][co][r][q] = w[co][ci][k-1-r][k-1-q], stride 1,
    pad k-1-pad (= pad for the 'same' convolutions of this model); stride-2 layers run on the generic / LDS-DMA kernels
    (zero-dilated gather), stride-1 3x3 layers may take the halo kernel.  Device-resident fp32 weights (training) are
    transposed, flipped, cast and laid out by one ``dy_pack_conv_weights`` launch."""
    k = weight.shape[2]
    # the streaming 1x1 kernel has no residual (accumulate) input: 1x1 layers take it only when the caller adds nothing in the epilogue
    halo = None if (stride == 1 and (k == 3 or (k == 1 and no_accumulate))) else False
    if weight.is_cuda and weight.dtype == torch.float32 and dtype != FP8 and weight.device == torch.device(device):
        return PackedConv(weight, zero_bias(weight.shape[1], weight.device), 1, k // 2, 1, False, dtype, device, halo=halo, transpose_flip=True)
    wt = weight.detach().permute(1, 0, 2, 3)  # a view: the packing copy below does the layout change
    if k > 1:
        wt = wt.flip(2, 3)
    return PackedConv(wt, zero_bias(wt.shape[0], wt.device), 1, k // 2, 1, False, dtype, device, halo=halo)


def conv_dgrad(dz: torch.Tensor, pc: PackedConv, stride: int, out: Optional[torch.Tensor] = None,
               accumulate: Optional[torch.Tensor] = None, bn_behind: Optional[BnBehind] = None) -> torch.Tensor:
    """dx of z = conv2d(x, w, stride, k//2) given dz, with ``pc = pack_dgrad(w, stride, ...)``; ``accumulate``: a gradient
    already held for x (another consumer's contribution), added in the epilogue; ``bn_behind``: see ``BnBehind`` (x is the output of a
    BatchNorm + activation layer with no other consumer: the sums of its backward come out of this kernel's epilogue where built)."""
    if stride not in (1, 2):
        raise NotImplementedError("conv_dgrad: stride 1 or 2")
    if bn_behind is not None and (accumulate is not None or dz.dtype not in (torch.bfloat16, torch.float16)):
        bn_behind = None  # (its slots stay 0: the BatchNorm backward reduces by itself)
    return conv2d(dz, pc, out=out, residual=accumulate, dil2=(stride == 2), bn_behind=bn_behind)


# ---- small training-path ops + optimizer --------------------------------------------------------------------------


def upsample2x_bwd(g: torch.Tensor) -> torch.Tensor:
    n, c, h2, w2 = g.shape
    out = alloc_nhwc(n, c, h2 // 2, w2 // 2, g.dtype, g.device)
    (gp, ldg), (op, ldo) = view_params(g), view_params(out)
    _launch(lib().dy_upsample2x_bwd_nhwc, (gp, op, n, h2 // 2, w2 // 2, c, ldg, ldo, dy_dtype(g.dtype)), keep=(g, out))
    return out


def maxpool_bwd(x: torch.Tensor, g_out: torch.Tensor, g_in: torch.Tensor, k: int, accumulate: bool) -> torch.Tensor:
    """g_in (+)= gradient of max_pool2d(x, k, 1, k//2) given g_out; all NHWC views of one dtype."""
    n, c, h, w = x.shape
    (xp, ldx), (gop, ldgo), (gip, ldgi) = view_params(x), view_params(g_out), view_params(g_in)
    _launch(lib().dy_maxpool_bwd_nhwc, (xp, gop, gip, n, h, w, c, ldx, ldgo, ldgi, k, int(accumulate), dy_dtype(x.dtype)), keep=(x, g_out, g_in))
    return g_in


def add_dilated2_(dx: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """dx[:, :, ::2, ::2] += t in place (``dy_add_dilated2_nhwc``): NHWC views of one dtype, t (n, c, ceil(H/2), ceil(W/2))."""
    n, c, h2, w2 = dx.shape
    if t.shape[0] != n or t.shape[1] != c or t.shape[2] != (h2 + 1) // 2 or t.shape[3] != (w2 + 1) // 2 or t.dtype != dx.dtype:
        raise ValueError("add_dilated2_: t must be the half-resolution map of dx")
    (tp, ldt), (xp, ldx) = view_params(t), view_params(dx)
    _launch(lib().dy_add_dilated2_nhwc, (tp, xp, n, t.shape[2], t.shape[3], h2, w2, c, ldt, ldx, dy_dtype(dx.dtype)), keep=(t, dx))
    return dx


def head_grad_split(g: torch.Tensor, nb: int, nc: int, ncp: int, dtype: torch.dtype, scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dzb, dzc) of ``dy_head_grad_split``: ``g`` an fp32 NHWC view (n, nb + nc, h, w); ``scale`` a device fp32 scalar or None."""
    n, c, h, w = g.shape
    if g.dtype != torch.float32 or c != nb + nc or g.stride(1) != 1:
        raise ValueError("head_grad_split: g must be an fp32 NHWC view of nb + nc channels")
    gp, ldg = view_params(g)
    dzb, dzc = alloc_nhwc(n, nb, h, w, dtype, g.device), alloc_nhwc(n, ncp, h, w, dtype, g.device)
    (bp, ldb), (cp, ldc) = view_params(dzb), view_params(dzc)
    if scale is not None and (scale.dtype != torch.float32 or not scale.is_cuda or scale.numel() != 1):
        raise ValueError("head_grad_split: scale must be one fp32 value on the device")
    _launch(lib().dy_head_grad_split, (gp, ldg, _rows(g), nb, nc, ncp, scale.data_ptr() if scale is not None else None, bp, ldb, cp, ldc, dy_dtype(dtype)),
            keep=(g, dzb, dzc, scale))
    return dzb, dzc


def add_nhwc(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, c, h, w = a.shape
    if out is None:
        out = alloc_nhwc(n, c, h, w, a.dtype, a.device)
    (ap, lda), (bp, ldb), (op, ldo) = view_params(a), view_params(b), view_params(out)
    _launch(lib().dy_add_nhwc, (ap, bp, op, _rows(a), c, lda, ldb, ldo, dy_dtype(a.dtype)), keep=(a, b, out))
    return out


def sumsq_into(acc: torch.Tensor, g: torch.Tensor) -> None:
    """acc (device double scalar) += sum(g^2); g: contiguous fp32."""
    _launch(lib().dy_sumsq_f32, (g.data_ptr(), g.numel(), acc.data_ptr()), keep=(acc, g))


def sgd_step_(p: torch.Tensor, grad: torch.Tensor, buf: torch.Tensor, lr: float, momentum: float, weight_decay: float, nesterov: bool,
              first_step: bool, grad_sumsq: Optional[torch.Tensor] = None, max_norm: float = 10.0, amp_state: Optional[torch.Tensor] = None) -> None:
    _launch(lib().dy_sgd_step, (p.data_ptr(), grad.data_ptr(), buf.data_ptr(), p.numel(), lr, momentum, weight_decay, int(nesterov), int(first_step),
                                grad_sumsq.data_ptr() if grad_sumsq is not None else None, max_norm, amp_state.data_ptr() if amp_state is not None else None),
            keep=(p, grad, buf, grad_sumsq, amp_state))


def adamw_step_(p: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, betas, eps: float, weight_decay: float, step: int,
                grad_sumsq: Optional[torch.Tensor] = None, max_norm: float = 10.0, amp_state: Optional[torch.Tensor] = None) -> None:
    _launch(lib().dy_adamw_step, (p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step,
                                  grad_sumsq.data_ptr() if grad_sumsq is not None else None, max_norm, amp_state.data_ptr() if amp_state is not None else None),
            keep=(p, grad, m, v, grad_sumsq, amp_state))


def amp_update_(amp_state: torch.Tensor, grad_sumsq: torch.Tensor, growth: float = 2.0, backoff: float = 0.5, interval: int = 2000) -> None:
    """``GradScaler.update()`` on the device state {scale, growth tracker, found_inf, skipped} (fp32[4]); no host synchronisation."""
    assert amp_state.dtype == torch.float32 and amp_state.numel() >= 4 and grad_sumsq.dtype == torch.float64
    _launch(lib().dy_amp_update, (amp_state.data_ptr(), grad_sumsq.data_ptr(), growth, backoff, interval), keep=(amp_state, grad_sumsq))


def ema_update_(ema: torch.Tensor, p: torch.Tensor, decay: float) -> None:
    _launch(lib().dy_ema_update, (ema.data_ptr(), p.data_ptr(), p.numel(), decay), keep=(ema, p))


def grad_sink_flush_(entries: torch.Tensor, grad: torch.Tensor, sink: torch.Tensor) -> None:
    """grad += sink, sink = 0 for every parameter block of ``entries`` (device int64 (n, 4): offset, cout, cin, kk) in one
    ``dy_grad_sink_flush`` launch; conv-weight blocks of ``sink`` are in the weight-gradient kernels' (cout, k, k, cin) order."""
    _launch(lib().dy_grad_sink_flush, (entries.data_ptr(), entries.shape[0], grad.data_ptr(), sink.data_ptr()), keep=(entries, grad, sink))


# ---- image sources ------------------------------------------------------------------------------------------------------


def letterbox(frames: torch.Tensor, new_w: int, new_h: int, top: int, left: int, hn: int, wn: int, swap_rb: bool = True,
              pad_value: float = 114.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 (N, H, W, 3) frames -> fp32 NCHW (N, 3, hn, wn) / 255 with the resized image at (top, left), 114 elsewhere."""
    require_device(frames, "frames")
    n, h0, w0, _ = frames.shape
    if out is None:
        out = torch.empty((n, 3, hn, wn), dtype=torch.float32, device=frames.device)
    _launch(lib().dy_letterbox_u8_to_nchw_f32, (frames.data_ptr(), out.data_ptr(), n, h0, w0, new_w, new_h, top, left, hn, wn, int(swap_rb), pad_value),
            keep=(frames, out))
    return out


def resize_bilinear_u8(img: torch.Tensor, size) -> torch.Tensor:
    """uint8 NCHW batch -> fp32 NCHW of ``size`` = interpolate(img.float() / 255, size, mode="bilinear", align_corners=False): the
    ``multi_scale`` branch of preprocess_batch (models/yolo/detect/train.py:60-73) in one kernel."""
    require_device(img, "image batch")
    if img.dtype != torch.uint8 or img.dim() != 4 or not img.is_contiguous():
        raise ValueError("resize_bilinear_u8 expects a contiguous uint8 (N, C, H, W) device tensor")
    n, c, h, w = img.shape
    ho, wo = int(size[0]), int(size[1])
    out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=img.device)
    _launch(lib().dy_resize_bilinear_u8_nchw_f32, (img.data_ptr(), out.data_ptr(), n, c, h, w, ho, wo), keep=(img, out))
    return out


def scale_img(img: torch.Tensor, ratio: float, gs: int = 32, flip_lr: bool = False) -> torch.Tensor:
    """``scale_img(img.flip(3) if flip_lr else img, ratio, gs=gs)`` of the reference (utils/torch_utils.py:436-445) in one kernel: bilinear resize to
    (int(h r), int(w r)), padded right / bottom with 0.447 to the next multiples of ``gs`` of (h r, w r) — the image pyramid of test-time augmentation."""
    import math

    require_device(img, "image batch")
    if img.dtype != torch.float32 or img.dim() != 4 or not img.is_contiguous():
        raise ValueError("scale_img expects a contiguous fp32 (N, C, H, W) device tensor")
    n, c, h, w = img.shape
    if ratio == 1.0 and not flip_lr:
        return img
    hs, ws = (int(h * ratio), int(w * ratio)) if ratio != 1.0 else (h, w)
    ho, wo = ((math.ceil(h * ratio / gs) * gs, math.ceil(w * ratio / gs) * gs) if ratio != 1.0 else (h, w))
    out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=img.device)
    _launch(lib().dy_scale_img_nchw_f32, (img.data_ptr(), out.data_ptr(), n, c, h, w, hs, ws, ho, wo, int(flip_lr), 0.447), keep=(img, out))
    return out


# ---- fused C2f block ------------------------------------------------------------------------------------------------------


class PackedC2f:
    """Folded + packed weights of a whole C2f (n = 1) for ``dy_c2f_fused``: (w, b) pairs of cv1, m[0].cv1, m[0].cv2, cv2."""

    def __init__(self, cv1, mcv1, mcv2, cv2, shortcut: bool, dtype: torch.dtype, device, act_l2e: bool = False):
        """``act_l2e``: the four (w, b) pairs were folded for the log2(e)-scaled activation domain (biases times log2 e)."""
        (w1, b1), (wa, ba), (wb, bb), (w2, b2) = cv1, mcv1, mcv2, cv2
        self.act_l2e = bool(act_l2e)
        self.cin, self.hidden, self.cout = w1.shape[1], wa.shape[0], w2.shape[0]
        self.w1, _ = pack_frag1x1(w1, b1, dtype, device)
        self.w2, _ = pack_frag1x1(w2, b2, dtype, device)
        pa = PackedConv(wa, ba, 1, 1, 1, True, dtype, device, halo=True)
        pb = PackedConv(wb, bb, 1, 1, 1, True, dtype, device, halo=True)
        if pa.layout != _lib.DY_WLAYOUT_HALO3X3 or pb.layout != _lib.DY_WLAYOUT_HALO3X3:
            raise ValueError("PackedC2f: the Bottleneck convolutions did not pack as DY_WLAYOUT_HALO3X3")
        self.wa, self.wb = pa.w, pb.w
        self.bias = torch.cat([t.detach().float().reshape(-1).cpu() for t in (b1, ba, bb, b2)]).contiguous().to(device)
        self.shortcut, self.dtype = bool(shortcut), dtype


def c2f_fused_supported(cin: int, hidden: int, cout: int, n: int, dtype: torch.dtype, cin_lo: int = 0) -> bool:
    """``cin`` counts every input channel of cv1; ``cin_lo`` of them come through a fused 2x upsample (the folded Concat)."""
    return dtype != FP8 and bool(lib().dy_c2f_fused_supported(cin, cin_lo, hidden, cout, n, dy_dtype(dtype)))


def c2f_fused(x: torch.Tensor, pk: PackedC2f, out: Optional[torch.Tensor] = None, x_lo: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Whole C2f block in one ``dy_c2f_fused`` launch; x: NHWC view (N, c, H, W) of pk.dtype.  With ``x_lo`` (N, c_lo, H/2, W/2) the
    block input is cat(upsample2x(x_lo), x) — the Upsample + Concat in front of the neck's C2f — and pk.cin = c_lo + c."""
    require_device(x, "c2f input")
    n, c, h, w = x.shape
    c_lo = 0
    if x_lo is not None:
        require_device(x_lo, "c2f upsampled input")
        c_lo = x_lo.shape[1]
        if x_lo.shape[0] != n or (2 * x_lo.shape[2], 2 * x_lo.shape[3]) != (h, w) or x_lo.dtype != x.dtype:
            raise ValueError("c2f_fused: x_lo must be the half-resolution map of the same batch and dtype")
    if c + c_lo != pk.cin or x.dtype != pk.dtype:
        raise ValueError("c2f_fused: input channels / dtype do not match the packed block")
    if out is None:
        out = alloc_nhwc(n, pk.cout, h, w, x.dtype, x.device)
    d = C2fDesc()
    (d.x, d.ld_x), (d.y, d.ld_y) = view_params(x), view_params(out)
    if x_lo is not None:
        d.x_lo, d.ld_x_lo = view_params(x_lo)
    d.w_cv1, d.w_m_cv1, d.w_m_cv2, d.w_cv2, d.bias = pk.w1.data_ptr(), pk.wa.data_ptr(), pk.wb.data_ptr(), pk.w2.data_ptr(), pk.bias.data_ptr()
    d.batch, d.h, d.w, d.cin, d.cin_lo, d.hidden, d.cout = n, h, w, pk.cin, c_lo, pk.hidden, pk.cout
    d.shortcut, d.dtype, d.act_l2e = int(pk.shortcut), dy_dtype(x.dtype), int(pk.act_l2e)
    _launch(lib().dy_c2f_fused, (C.byref(d),), keep=(d, x, x_lo, out, pk))
    return out
