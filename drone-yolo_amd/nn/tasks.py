"""Model graph: YAML -> modules -> executor (reference: ultralytics/nn/tasks.py).

``yaml_model_load`` (:1093-1124), ``guess_model_scale`` (:1127-1141), ``parse_model`` (:929-1090),
``BaseModel`` (:95-295) and ``DetectionModel`` (:299-345) keep the reference's names and
arguments.  What differs is MI355X-first:

* ``RepVGGBlock`` is a first-class base module here.  In the reference it is exported
  (nn/modules/__init__.py:62) but missing from ``parse_model``'s globals and ``base_modules``
  (tasks.py:12-66, 954-991), so the shipped YAML cannot be built from scratch there.
* the executor plans Concat by construction: a layer whose output feeds a Concat writes into its
  channel slice of that Concat's buffer (``out=``), and Upsample+Concat in front of a C2f are
  folded into the C2f.cv1 gather (dual-source + 2x-nearest addressing in ``dy_conv2d_nhwc``).
* strides are derived from the graph instead of a CPU forward on zeros (tasks.py:324-337): the
  product path has no CPU forward.
* a forward is recorded once per input shape as a ``LaunchPlan`` and replayed afterwards.
"""
from __future__ import annotations

import ast
import contextlib
import math
import os
import re
from copy import deepcopy
from pathlib import Path
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import yaml

from .. import hip_ops as H
from ..utils import LOGGER
from ..utils.ops import make_divisible
from ..utils.torch_utils import initialize_weights
from .modules import SPPF, Bottleneck, C2f, Concat, Conv, Detect, DWConv, RepVGGBlock, Upsample
from .modules.block import DFL
from .modules.conv import PlainConv2d

CFG_DIR = Path(__file__).resolve().parents[1] / "cfg"

_MODULES = {
    "Conv": Conv, "DWConv": DWConv, "RepVGGBlock": RepVGGBlock, "C2f": C2f, "SPPF": SPPF, "Bottleneck": Bottleneck,
    "Concat": Concat, "Detect": Detect, "nn.Upsample": Upsample,
}  # fmt: skip
_BASE_MODULES = frozenset({Conv, DWConv, RepVGGBlock, C2f, SPPF, Bottleneck})
_REPEAT_MODULES = frozenset({C2f})


def guess_model_scale(model_path) -> str:
    """'yolov8s-p2-repvgg.yaml' -> 's' (reference tasks.py:1127-1141)."""
    with contextlib.suppress(AttributeError):
        return re.search(r"yolo[v]?\d+([nslmx])", Path(model_path).stem).group(1)
    return ""


def yaml_model_load(path) -> dict:
    """Load a model YAML; the scale letter in the file name selects ``scales[...]`` (tasks.py:1093-1124)."""
    path = Path(path)
    unified = re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", str(path))  # yolov8s-p2.yaml -> yolov8-p2.yaml
    for cand in (Path(unified), path, CFG_DIR / "models" / "v8" / Path(unified).name, CFG_DIR / "models" / "v8" / path.name):
        if cand.is_file():
            with open(cand, errors="ignore", encoding="utf-8") as f:
                d = yaml.safe_load(f)
            d["scale"] = guess_model_scale(path)
            d["yaml_file"] = str(path)
            return d
    raise FileNotFoundError(f"model YAML '{path}' not found (also looked in {CFG_DIR / 'models' / 'v8'})")


def parse_model(d: dict, ch: int, verbose: bool = True):
    """YAML dict -> (nn.Sequential of layers, save list); same walk as reference tasks.py:929-1090."""
    legacy = True  # v8 YAMLs: Detect keeps the two-3x3 class branch (tasks.py:934,1062)
    max_channels = float("inf")
    nc, act, scales = (d.get(x) for x in ("nc", "activation", "scales"))
    depth, width = (d.get(x, 1.0) for x in ("depth_multiple", "width_multiple"))
    if scales:
        scale = d.get("scale")
        if not scale:
            scale = tuple(scales.keys())[0]
            LOGGER.warning(f"WARNING no model scale passed. Assuming scale='{scale}'.")
        depth, width, max_channels = scales[scale]
    if act:
        raise NotImplementedError("custom 'activation:' in the YAML is not supported: the conv epilogue builds SiLU only")
    if verbose:
        LOGGER.info(f"\n{'':>3}{'from':>20}{'n':>3}{'params':>10}  {'module':<45}{'arguments':<30}")
    ch = [ch]
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, mname, args) in enumerate(d["backbone"] + d["head"]):
        if mname not in _MODULES:
            raise NotImplementedError(f"module '{mname}' (layer {i}) is not on the Drone-YOLO detection path")
        m = _MODULES[mname]
        args = list(args)
        for j, a in enumerate(args):
            if isinstance(a, str):
                with contextlib.suppress(ValueError, SyntaxError):
                    args[j] = nc if a == "nc" else ast.literal_eval(a)
        n = n_ = max(round(n * depth), 1) if n > 1 else n
        if m in _BASE_MODULES:
            c1, c2 = ch[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if m in _REPEAT_MODULES:
                args.insert(2, n)
                n = 1
        elif m is Concat:
            c2 = sum(ch[x] for x in f)
        elif m is Detect:
            args.append([ch[x] for x in f])
            m.legacy = legacy
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        t = mname
        m_.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, t
        if verbose:
            LOGGER.info(f"{i:>3}{str(f):>20}{n_:>3}{m_.np:10.0f}  {t:<45}{str(args):<30}")
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


def _layer_stride(m: nn.Module) -> float:
    """Spatial scale factor of one top-level layer (input size / output size)."""
    if isinstance(m, nn.Sequential):
        return math.prod(_layer_stride(x) for x in m)
    if isinstance(m, Conv):
        return float(m.conv.stride[0])
    if isinstance(m, RepVGGBlock):
        return float(m.stride)
    if isinstance(m, Upsample):
        return 1.0 / float(m.scale_factor)
    return 1.0


class BaseModel(nn.Module):
    """Sequential executor with a skip list — reference tasks.py:95-295."""

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            return self.loss(x, *args, **kwargs)
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None):
        if profile or visualize or embed:
            raise NotImplementedError("profile/visualize/embed are outside the accelerated path")
        if augment:
            return self._predict_augment(x)
        return self._predict_once(x)

    def _predict_augment(self, x, image_dtype=None):
        """Test-time augmentation — reference DetectionModel._predict_augment / _descale_pred / _clip_augmented (nn/tasks.py:347-383): the batch at
        scales 1, 0.83 (mirrored left-right) and 0.67 (``dy_scale_img_nchw_f32``: resize + pad in one kernel), three passes of the path, every pass's
        boxes scaled and mirrored back into the input's frame, the coarsest level of the first pass and the finest level of the last one dropped, all
        anchors side by side: ((B, 4 + nc, A_total), None).  Eval mode, Detect returning its decoded output."""
        if self.training or not isinstance(self.model[-1], Detect):
            raise NotImplementedError("augment=True is built for the Detect models in eval mode")
        x = x.float().contiguous()
        img_size = x.shape[-2:]
        gs = int(self.stride.max())
        ys = []
        for si, fi in zip((1, 0.83, 0.67), (None, 3, None)):
            xi = H.scale_img(x, si, gs=gs, flip_lr=(fi == 3))
            yi = self._predict_once(xi, image_dtype=image_dtype)[0].clone()  # (the pass's output buffer belongs to its recorded / cached plan)
            yi[:, :4] /= si  # de-scale
            if fi == 3:
                yi[:, 0] = img_size[1] - yi[:, 0]  # de-flip lr
            ys.append(yi)
        nl = self.model[-1].nl
        g = sum(4 ** k for k in range(nl))  # grid points of the level pyramid relative to its coarsest level
        i = ys[0].shape[-1] // g  # large: without its coarsest level (the tail of the anchor axis)
        ys[0] = ys[0][..., :-i]
        i = (ys[-1].shape[-1] // g) * 4 ** (nl - 1)  # small: without its finest level (the head)
        ys[-1] = ys[-1][..., i:]
        return torch.cat(ys, -1), None

    # ---- graph planning ---------------------------------------------------------------------------
    def _plan_graph(self) -> None:
        """Decide, per layer, where its output lives (Concat by construction) and which
        Upsample/Concat pairs are folded into the consuming C2f."""
        layers = list(self.model)
        self._srcs = {m.i: [(m.i - 1) if s == -1 else (m.i + s if s < 0 else s) for s in
                            ([m.f] if isinstance(m.f, int) else list(m.f))] for m in layers}
        consumers: Dict[int, List[int]] = {i: [] for i in range(len(layers))}
        for i, srcs in self._srcs.items():
            for s in srcs:
                if i > 0:
                    consumers[s].append(i)
        self._consumers0 = list(consumers[0])
        self._virtual: Dict[int, tuple] = {}  # concat layer index -> (lowres_src, skip_src) folded into next C2f
        self._skip: set = set()  # layers that launch nothing (folded Upsample / Concat)
        self._place: Dict[int, tuple] = {}  # producer layer -> (concat layer, channel offset)
        fold = getattr(self, "fold_concat", True)
        for m in layers:
            if not isinstance(m, Concat):
                continue
            src = self._srcs[m.i]
            nxt = layers[m.i + 1] if m.i + 1 < len(layers) else None
            up = layers[src[0]]
            if (fold and len(src) == 2 and isinstance(up, Upsample) and consumers[up.i] == [m.i]
                    and isinstance(nxt, C2f) and consumers[m.i] == [nxt.i] and self._srcs[nxt.i] == [m.i]):
                self._virtual[m.i] = (self._srcs[up.i][0], src[1])
                self._skip.update((up.i, m.i))
                continue
            off = 0
            for s in src:
                c = self._out_ch[s]
                n_cat = sum(1 for k in consumers[s] if isinstance(layers[k], Concat) and k not in self._virtual)
                if s not in self._place and n_cat == 1 and not isinstance(layers[s], (Concat, Detect, nn.Sequential)):
                    self._place[s] = (m.i, off)
                off += c

    def _scaled_domain_for(self, dtype) -> bool:
        """Whether a pass in ``dtype`` runs in the log2(e)-scaled activation domain (hip_ops.scaled_activations): 16-bit and fp8 storage
        (fp32 keeps the reference's units: it is the bar-exact precision, not a throughput one), layer 0 a Conv / RepVGGBlock (it reads
        the raw image and is packed with its weights scaled), and nothing but modules known to commute with a positive scale."""
        ok = self.__dict__.get("_l2e_ok")
        if ok is None:
            ok = os.environ.get("DYOLO_L2E", "1") != "0" and isinstance(self.model[0], (Conv, RepVGGBlock)) and not isinstance(self.model[0], DWConv)
            known = (Conv, RepVGGBlock, C2f, SPPF, Bottleneck, Concat, Upsample, Detect, nn.Sequential, nn.ModuleList, PlainConv2d, nn.Conv2d, nn.BatchNorm2d,
                     nn.SiLU, nn.Identity, nn.MaxPool2d, DFL)
            ok = ok and all(isinstance(m, known) for m in self.model.modules())
            if ok:
                self.model[0]._raw_input = True
            self.__dict__["_l2e_ok"] = ok
        return ok and dtype in (torch.bfloat16, torch.float16, H.FP8)

    def _predict_once(self, x, image_dtype=None):
        with H.scaled_activations(self._scaled_domain_for(image_dtype if image_dtype is not None else x.dtype)):
            return self._predict_layers(x, image_dtype)

    def _predict_layers(self, x, image_dtype=None):
        """Run every layer.  ``x`` is an NHWC-view tensor (see hip_ops) or, with ``image_dtype`` set, the raw
        contiguous fp32 NCHW image: the first layer then runs the fused stem kernel (layout cast + conv) when
        its shape allows, otherwise the image is converted first."""
        if not hasattr(self, "_place"):
            self._plan_graph()
        stem_image = None
        if image_dtype is not None:
            first = self.model[0]
            if isinstance(first, Conv) and first.is_stem() and 0 not in self._place and getattr(self, "fuse_stem", True) and \
                    (image_dtype != H.F16X2 or (first.conv.out_channels % 8 == 0 and first.conv.out_channels <= 64)):
                stem_image = x
                n, _, h, w = x.shape
                x = torch.empty((n, 0, h, w), dtype=image_dtype, device=x.device)  # shape/dtype carrier only
            elif image_dtype == H.FP8:
                raise NotImplementedError("fp8 storage needs the image stem (Conv(3, c, 3, 2)) as the first layer: it runs in fp16 and hands over quantised")
            else:
                x = H.to_nhwc(x, image_dtype, mark_input=True)
        y: List[Optional[torch.Tensor]] = []
        cat_bufs: Dict[int, torch.Tensor] = {}
        fused_stem2 = None
        # mixed-precision plan (BASELINE config 5): the C2f blocks in ``fp8_layers`` run their INTERNALS in e4m3 on the
        # block-scaled fp8 MFMA — cv1 reads the 16-bit input and writes fp8, the Bottlenecks are fp8, cv2 reads fp8 and writes the 16-bit
        # output — so every layer boundary, skip connection and Concat stays 16-bit and no plan can leave a consumer with the wrong type
        f8 = self.__dict__.get("fp8_layers") or frozenset()
        if f8 and x.dtype != torch.float16:
            raise NotImplementedError("a mixed fp8 plan (fp8_layers) runs on float16 storage")
        for m in self.model:
            i = m.i
            if i in self._skip:
                y.append(None)
                continue
            src = self._srcs[i]
            kw = {}
            if isinstance(m, C2f) and src[0] in self._virtual:
                lo, skip = self._virtual[src[0]]
                xin, kw = y[lo], {"x2": y[skip], "up2x": True}
            elif isinstance(m.f, int):
                xin = x if (i == 0) else y[src[0]]
            else:
                xin = [y[j] for j in src]
            if i in f8:
                if not isinstance(m, C2f) or kw:
                    raise NotImplementedError(f"fp8_layers: layer {i} ({type(m).__name__}) — fp8 internals are built for C2f blocks with one plain input")
                kw["fp8_internal"] = True
            if i in self._place:
                ci, off = self._place[i]
                if ci not in cat_bufs:
                    n = x.shape[0]
                    h, w = self._out_hw(ci, x.shape[2], x.shape[3])
                    cat_bufs[ci] = H.alloc_nhwc(n, self._out_ch[ci], h, w, x.dtype, x.device)
                kw["out"] = cat_bufs[ci][:, off : off + self._out_ch[i]]
            if isinstance(m, Concat) and i in cat_bufs:
                kw["out"] = cat_bufs[i]
            if i == 0 and stem_image is not None:
                pk2 = self._stem2_pack(stem_image, image_dtype, consumers0=self._consumers0)
                if pk2 is not None:  # layers 0 + 1 in one kernel: the half-resolution map never reaches HBM
                    y.append(None)
                    with H.layer_tag((0, "Conv + RepVGGBlock (layers 0 + 1, fused)")):
                        fused_stem2 = H.stem2_fused(stem_image, pk2, mark_input=True)
                    continue
                with H.layer_tag((0, type(m).__name__)):
                    if image_dtype == H.FP8:  # the 3-channel image layer runs in fp16 (K = 27), its output is quantised once
                        out = H.quantize_fp8(m.forward_stem(stem_image, torch.float16, mark_input=True), out=kw.get("out"))
                    else:
                        out = m.forward_stem(stem_image, image_dtype, mark_input=True, **kw)
            elif i == 1 and fused_stem2 is not None:
                out = fused_stem2
            else:
                with H.layer_tag((i, type(m).__name__)):
                    out = m(xin, **kw)
            y.append(out)
        return y[-1]

    fuse_stem2 = True  # layers 0 + 1 in one kernel where dy_stem2_fused is built for the shapes

    def fp8_plan_off_p2(self) -> frozenset:
        """The mixed plan the error budget allows on these models (profiles/r04_fp8_sensitivity_*.jsonl): the C2f blocks that do NOT feed
        the highest-resolution Detect level.  Most detections of the P2 models live on that level (25,600 of 34,000 anchors at 640 x
        640), and ONE e4m3 rounding anywhere on its path flips 7-15 % of the kept boxes of the synthetic-weight fixtures, while the deep
        neck blocks (stride 8 / 16 / 32 outputs) in fp8 flip < 1 %."""
        if not hasattr(self, "_srcs"):
            self._plan_graph()
        det = self.model[-1]
        if not isinstance(det, Detect):
            return frozenset()
        anc, todo = set(), [self._srcs[det.i][0]]
        while todo:
            t = todo.pop()
            if t in anc:
                continue
            anc.add(t)
            if t > 0:
                todo += list(self._srcs[t])
        return frozenset(m.i for m in self.model if isinstance(m, C2f) and m.i not in anc and m.i not in self._skip
                         and not (self._srcs[m.i][0] in self._virtual))

    def _stem2_pack(self, image, dtype, consumers0):
        """Packed weights for ``dy_stem2_fused`` when layers 0 / 1 are Conv(3, 32, 3, 2) -> RepVGGBlock(32, 64, stride 2),
        layer 0 feeds nothing else and neither output is a Concat slice; else None (layer-by-layer path)."""
        if not self.fuse_stem2 or len(self.model) < 2 or consumers0 != [1] or 0 in self._place or 1 in self._place:
            return None
        m0, m1 = self.model[0], self.model[1]
        if not (isinstance(m1, RepVGGBlock) and isinstance(m1.se, nn.Identity) and m1.stride == 2 and m1.groups == 1
                and m0.conv.out_channels == 32 and m1.in_channels == 32 and isinstance(m0.act, nn.SiLU)):
            return None
        c1 = (m1.rbr_reparam if hasattr(m1, "rbr_reparam") else m1.rbr_dense.conv).out_channels
        n, c, h, w = image.shape
        if not H.stem2_fused_supported(c, 32, c1, h, w, dtype):
            return None
        srcs = [m0.conv.weight, m0.bn.weight, m0.bn.bias, m0.bn.running_mean, m0.bn.running_var] + list(m1.parameters()) + list(m1.buffers())
        key = (dtype, str(image.device), H.scaled_domain(), tuple((t.data_ptr(), t._version) for t in srcs))
        cache = self.__dict__.get("_stem2_cache")
        if cache is None or cache[0] != key:
            from .modules.conv import fold_conv_bn

            w0, b0 = fold_conv_bn(m0.conv.weight, m0.conv.bias, m0.bn)
            w1, b1 = (m1.rbr_reparam.weight, m1.rbr_reparam.bias) if hasattr(m1, "rbr_reparam") else m1.get_equivalent_kernel_bias()
            w0, b0, a0 = H.domain_fold(w0, b0, True, raw_input=True)  # the image is raw; layer 1 reads layer 0's (scaled) output
            w1, b1, a1 = H.domain_fold(w1, b1, True)
            cache = (key, H.PackedStem2(w0, b0, a0, w1, b1, a1, dtype, image.device))
            self.__dict__["_stem2_cache"] = cache
        return cache[1]

    def _out_hw(self, i: int, h: int, w: int):
        s = self._cum_stride[i]
        return int(round(h / s)), int(round(w / s))

    _PACK_CACHES = ("_packed", "_block_cache", "_tail_cache", "_first_cache", "_stem2_cache")

    def drop_packed(self) -> None:
        """Forget every packed (BatchNorm-folded, device-layout) copy of the weights and advance the weights epoch that
        recorded launch plans are keyed on (``weights_signature``)."""
        self.__dict__["_weights_epoch"] = self.__dict__.get("_weights_epoch", 0) + 1
        self.__dict__.pop("_sig_tensors", None)
        for m in self.modules():
            for k in self._PACK_CACHES:
                m.__dict__.pop(k, None)

    def weights_signature(self):
        """Changes whenever the weights a recorded pass baked in may have changed: storage identity and torch's in-place
        version counter of every parameter / buffer (load_state_dict, .to(), in-place edits) plus the weights epoch
        (train() <-> eval() transitions: the trainer's kernels write parameters through raw pointers, which torch's
        version counters cannot see)."""
        ts = self.__dict__.get("_sig_tensors")
        if ts is None:  # the module walk costs ~1 ms; the cached list ~60 us per pass
            ts = self.__dict__["_sig_tensors"] = list(self.parameters()) + list(self.buffers())
        return (self.__dict__.get("_weights_epoch", 0), tuple((t.data_ptr(), t._version) for t in ts))

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_sig_tensors", None)
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode: bool = True):
        changed = mode != self.training
        super().train(mode)
        if changed:
            self.drop_packed()
        return self

    def fuse(self, verbose=True):
        """Reference tasks.py:193-221 folds Conv+BN in place; here folding happens when weights are
        packed for the device, so this only drops stale packs and reports."""
        self.drop_packed()
        return self

    def is_fused(self, thresh=10):
        return True

    def info(self, detailed=False, verbose=True, imgsz=640):
        n_p = sum(p.numel() for p in self.parameters())
        n_l = len(list(self.modules()))
        if verbose:
            LOGGER.info(f"{Path(self.yaml.get('yaml_file', 'model')).stem} summary: {n_l} modules, {n_p:,} parameters")
        return n_l, n_p

    def load(self, weights, verbose=True):
        """Load a state dict (or a module) with matching keys/shapes — reference tasks.py:265-278."""
        sd = weights.state_dict() if isinstance(weights, nn.Module) else (weights.get("model", weights) if isinstance(
            weights, dict) else weights)
        if isinstance(sd, nn.Module):
            sd = sd.float().state_dict()
        own = self.state_dict()
        ok = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
        self.load_state_dict(ok, strict=False)
        if verbose:
            LOGGER.info(f"Transferred {len(ok)}/{len(own)} items from pretrained weights")

    train_dtype = torch.bfloat16  # storage type of activations in training (the reference trains under AMP, trainer.py:379)

    def forward_train(self, img, dtype=None):
        """Training-mode forward (batch-statistics BatchNorm, autograd graph): Detect's raw per-level maps."""
        from .train_forward import model_train_forward

        return model_train_forward(self, img, dtype or self.train_dtype)

    def loss(self, batch, preds=None):
        """criterion(model(batch['img']), batch) — reference tasks.py:280-292."""
        if getattr(self, "criterion", None) is None:
            self.criterion = self.init_criterion()
        preds = self.forward_train(batch["img"]) if preds is None else preds
        return self.criterion(preds, batch)


class DetectionModel(BaseModel):
    """YOLO detection model — reference tasks.py:299-345."""

    def __init__(self, cfg="yolov8s-p2-repvgg.yaml", ch=3, nc=None, verbose=True):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            if verbose:
                LOGGER.info(f"Overriding model.yaml nc={self.yaml['nc']} with nc={nc}")
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self.names = {i: f"{i}" for i in range(self.yaml["nc"])}
        self.inplace = self.yaml.get("inplace", True)
        self.end2end = False

        # per-layer channels and cumulative stride (replaces the zeros(1,ch,256,256) probe, tasks.py:324-337)
        self._out_ch: Dict[int, int] = {}
        self._cum_stride: Dict[int, float] = {}
        for m in self.model:
            src = [m.f] if isinstance(m.f, int) else list(m.f)
            src = [(m.i - 1) if s == -1 else (m.i + s if s < 0 else s) for s in src]
            s_in = 1.0 if m.i == 0 else self._cum_stride[src[0]]
            self._cum_stride[m.i] = s_in * _layer_stride(m)
            if isinstance(m, Concat):
                self._out_ch[m.i] = sum(self._out_ch[s] for s in src)
            elif isinstance(m, (Upsample,)):
                self._out_ch[m.i] = self._out_ch[src[0]]
            elif isinstance(m, Detect):
                self._out_ch[m.i] = m.no
            else:
                last = m[-1] if isinstance(m, nn.Sequential) else m
                self._out_ch[m.i] = _module_out_channels(last)
        det = self.model[-1]
        if isinstance(det, Detect):
            src = [(det.i + s if s < 0 else s) for s in det.f]
            det.stride = torch.tensor([self._cum_stride[s] for s in src])
            self.stride = det.stride
            det.bias_init()
        else:
            self.stride = torch.Tensor([32])
        initialize_weights(self)
        if verbose:
            self.info()

    def init_criterion(self):
        from ..utils.loss import v8DetectionLoss

        return v8DetectionLoss(self)


def _module_out_channels(m: nn.Module) -> int:
    if isinstance(m, Conv):
        return m.conv.out_channels
    if isinstance(m, RepVGGBlock):
        return (m.rbr_reparam if hasattr(m, "rbr_reparam") else m.rbr_dense.conv).out_channels
    if isinstance(m, (C2f, SPPF)):
        return m.cv2.conv.out_channels
    if isinstance(m, Bottleneck):
        return m.cv2.conv.out_channels
    raise TypeError(f"cannot infer output channels of {type(m).__name__}")
