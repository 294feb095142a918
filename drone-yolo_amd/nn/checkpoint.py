"""Reading the reference's pickled ``.pt`` checkpoints without the ``ultralytics`` package.

Reference: ``torch_safe_load`` / ``attempt_load_one_weight`` (ultralytics/nn/tasks.py:786-926) unpickle a dict whose
``"ema"`` / ``"model"`` entries are whole ``ultralytics.nn.tasks.DetectionModel`` module graphs (saved fp16 by
``BaseTrainer.save_model``, engine/trainer.py:514-545) — which needs the ultralytics classes importable.  Here a restricted
unpickler resolves every ``ultralytics.*`` class name to the same-named class of this package when it has one (Conv,
C2f, SPPF, RepVGGBlock, Detect, DetectionModel ... keep the reference's attribute and child names, so the restored objects
are ordinary ``nn.Module``s whose ``state_dict()`` has the reference's keys) and to an inert placeholder otherwise
(trainer arguments, loss objects, callbacks).  Globals are resolved through an EXACT (module, name) allow-list — tensor /
storage / parameter rebuild helpers, ``OrderedDict``, numpy array reconstruction, ``torch.nn.modules.*`` layer classes and
a few builtin containers; dotted names (``torch`` + ``serialization.os.system``) are refused, as is everything else.

The result is turned into THIS package's ``DetectionModel`` built from the pickled model's ``yaml`` dict and loaded with
the pickled weights (fp32); BatchNorm eps / momentum are set by ``initialize_weights`` as in the reference.
"""
from __future__ import annotations

import pickle
import types
from typing import Any, Dict, Tuple

import torch
import torch.nn as nn


class _Placeholder:
    """Stands in for reference classes this package has no use for (their state is kept but inert)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})

    def __call__(self, *a, **k):
        return self


# EXACT (module, name) pairs that may be resolved while reading a checkpoint.  A prefix allow-list is not enough: pickle
# resolves dotted names attribute by attribute, so ("torch", "serialization.os.system") would walk out of torch; every
# name with a "." in it is refused outright, and nothing callable beyond tensor / container reconstruction is reachable.
_TORCH_STORAGES = ("FloatStorage", "HalfStorage", "BFloat16Storage", "DoubleStorage", "LongStorage", "IntStorage", "ShortStorage", "CharStorage",
                   "ByteStorage", "BoolStorage")
_TORCH_DTYPES = ("float16", "float32", "float64", "bfloat16", "int8", "uint8", "int16", "int32", "int64", "bool", "half", "float", "double", "long", "int")
_EXACT = {("collections", "OrderedDict"), ("torch", "Size"), ("torch", "device"),
          ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
          ("torch._utils", "_rebuild_tensor"), ("torch.nn.parameter", "Parameter"),
          ("numpy", "ndarray"), ("numpy", "dtype"),
          ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
          ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
          ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"), ("_codecs", "encode")}
_EXACT |= {("torch", n) for n in _TORCH_STORAGES} | {("torch", n) for n in _TORCH_DTYPES}
_INERT = {("argparse", "Namespace"), ("pathlib", "WindowsPath"), ("pathlib", "PureWindowsPath")}  # restored as placeholders, never instantiated
_BUILTINS_OK = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice", "range", "object"}


def _own_classes() -> Dict[str, type]:
    from . import modules as M
    from . import tasks as T

    out = {n: getattr(M, n) for n in dir(M) if isinstance(getattr(M, n), type)}
    out["DetectionModel"] = T.DetectionModel
    out["BaseModel"] = T.BaseModel
    return out


class RefUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if "." in name or not name.isidentifier():
            raise pickle.UnpicklingError(f"refusing dotted / malformed global {module}.{name} in a checkpoint")
        if module.startswith("ultralytics.") or module == "ultralytics" or module.startswith("models.") or module == "models":
            return _own_classes().get(name, type(name, (_Placeholder,), {}))
        if module in ("builtins", "__builtin__"):  # torch.save writes protocol-2 pickles: builtins appear as __builtin__
            if name not in _BUILTINS_OK:
                raise pickle.UnpicklingError(f"refusing builtins.{name} in a checkpoint")
            return super().find_class("builtins", name)
        if (module, name) in _INERT:
            return type(name, (_Placeholder,), {})
        if (module, name) in _EXACT:
            return super().find_class(module, name)
        if module.startswith("torch.nn.modules."):  # layer classes of the pickled module graph (Conv2d, BatchNorm2d, SiLU, Sequential ...)
            obj = super().find_class(module, name)
            if isinstance(obj, type) and issubclass(obj, nn.Module):
                return obj
        raise pickle.UnpicklingError(f"refusing to import {module}.{name} while reading a checkpoint")


def _pickle_module():
    m = types.ModuleType("dyolo_ref_pickle")
    m.Unpickler = RefUnpickler
    m.load = lambda f, **kw: RefUnpickler(f, **kw).load()
    m.__name__ = "pickle"  # torch.load inspects only Unpickler / load
    return m


def read_reference_checkpoint(path: str) -> Tuple[dict, Dict[str, torch.Tensor], Dict[str, Any]]:
    """(model yaml dict, fp32 state dict with the reference's keys, remaining checkpoint entries)."""
    ckpt = torch.load(path, map_location="cpu", pickle_module=_pickle_module(), weights_only=False)
    if isinstance(ckpt, dict) and "yaml" in ckpt and isinstance(ckpt.get("model"), dict):
        return ckpt["yaml"], {k: v.float() if v.is_floating_point() else v for k, v in ckpt["model"].items()}, {}
    mod = (ckpt.get("ema") or ckpt["model"]) if isinstance(ckpt, dict) else ckpt  # tasks.py:906
    if not isinstance(mod, nn.Module):
        raise TypeError(f"{path}: no module under 'ema' / 'model' (got {type(mod).__name__})")
    yaml_d = getattr(mod, "yaml", None)
    if not isinstance(yaml_d, dict):
        raise ValueError(f"{path}: the pickled model carries no yaml dict")
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in mod.state_dict().items()}
    meta = {k: v for k, v in ckpt.items() if k not in ("model", "ema", "optimizer")} if isinstance(ckpt, dict) else {}
    names = getattr(mod, "names", None)
    if names is not None:
        meta["names"] = names
    return dict(yaml_d), sd, meta


def read_checkpoint_dict(path: str) -> Dict[str, Any]:
    """The whole checkpoint dictionary as written by ``BaseTrainer.save_model`` (``ema`` = the module graph restored as THIS package's
    classes, ``optimizer`` = torch.optim's state-dict layout, ``epoch``, ``best_fitness``, ``updates``, ``train_args`` ...) through the
    restricted unpickler: what ``resume_training`` and ``strip_optimizer`` work on (engine/trainer.py:731-754, utils/torch_utils.py:553-616)."""
    ckpt = torch.load(str(path), map_location="cpu", pickle_module=_pickle_module(), weights_only=False)
    if not isinstance(ckpt, dict):
        raise TypeError(f"{path}: checkpoint is not a Python dictionary")
    return ckpt


def load_reference_checkpoint(path: str, verbose: bool = False):
    """A ``DetectionModel`` of this package carrying the checkpoint's architecture and weights."""
    from .tasks import DetectionModel

    yaml_d, sd, meta = read_reference_checkpoint(path)
    model = DetectionModel(dict(yaml_d), ch=yaml_d.get("ch", 3), nc=yaml_d.get("nc"), verbose=verbose)
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    extra = [k for k in sd if k not in own]
    bad = [k for k in own if k in sd and tuple(own[k].shape) != tuple(sd[k].shape)]
    if missing or bad:
        raise ValueError(f"{path}: checkpoint does not fit the model built from its yaml (missing {missing[:3]}, shape mismatch {bad[:3]}, extra {extra[:3]})")
    model.load_state_dict({k: sd[k] for k in own})
    if isinstance(meta.get("names"), dict):
        model.names = meta["names"]
    return model, meta


# ---- writing: a checkpoint the reference itself can torch.load (engine/trainer.py:514-545) ------------------------------------
_REF_PATHS = {"Conv": "ultralytics.nn.modules.conv", "DWConv": "ultralytics.nn.modules.conv", "Concat": "ultralytics.nn.modules.conv",
              "DFL": "ultralytics.nn.modules.block", "SPPF": "ultralytics.nn.modules.block", "C2f": "ultralytics.nn.modules.block",
              "Bottleneck": "ultralytics.nn.modules.block", "RepVGGBlock": "ultralytics.nn.modules.block", "SEBlock": "ultralytics.nn.modules.block",
              "Detect": "ultralytics.nn.modules.head", "DetectionModel": "ultralytics.nn.tasks", "BaseModel": "ultralytics.nn.tasks"}
_DROP_ATTRS = ("_packed", "_block_cache", "_tail_cache", "_first_cache", "_stem2_cache", "_sig_tensors", "_weights_epoch", "_place", "_srcs", "_virtual", "_skip",
               "_consumers0", "_out_ch", "_cum_stride", "criterion", "train_dtype", "args", "fused_nms", "fuse_tail")


class _reference_class_paths:
    """While active, this package's module classes present themselves to ``pickle`` under the reference's import paths
    (``ultralytics.nn.modules.conv.Conv`` ...): throw-away entries in ``sys.modules`` that hold OUR classes, and the classes'
    ``__module__`` pointed at them, so ``save_global`` writes the reference's names.  Everything is restored on exit."""

    def __enter__(self):
        import sys

        self._saved_modules, self._saved_attr = {}, []
        own = _own_classes()
        for name, path in _REF_PATHS.items():
            cls = own.get(name)
            if cls is None:
                continue
            parts = path.split(".")
            for i in range(1, len(parts) + 1):
                mod_name = ".".join(parts[:i])
                if mod_name not in self._saved_modules:
                    self._saved_modules[mod_name] = sys.modules.get(mod_name)
                    if not isinstance(sys.modules.get(mod_name), types.ModuleType) or not getattr(sys.modules[mod_name], "__dyolo_fake__", False):
                        fake = types.ModuleType(mod_name)
                        fake.__dyolo_fake__ = True
                        sys.modules[mod_name] = fake
            setattr(sys.modules[path], name, cls)
            self._saved_attr.append((cls, cls.__module__, cls.__qualname__))
            cls.__module__, cls.__qualname__ = path, name
        return self

    def __exit__(self, *exc):
        import sys

        for cls, mod, qual in self._saved_attr:
            cls.__module__, cls.__qualname__ = mod, qual
        for mod_name, old in self._saved_modules.items():
            if old is None:
                sys.modules.pop(mod_name, None)
            else:
                sys.modules[mod_name] = old
        return False


def reference_module_graph(model: nn.Module, state_dict: Dict[str, torch.Tensor], half: bool = True) -> nn.Module:
    """A CPU deep copy of ``model`` carrying ``state_dict``, shaped the way the reference expects its own pickled module graph:
    caches and planner state dropped, the plain last conv of each Detect branch and the Upsample layers turned back into
    ``torch.nn`` instances, ``.type`` strings as the reference's ``parse_model`` writes them, fp16 like ``ema.half()``."""
    from copy import deepcopy

    from .modules.conv import PlainConv2d, Upsample

    live = {id(m): {k: m.__dict__.pop(k) for k in _DROP_ATTRS if k in m.__dict__} for m in model.modules()}  # do not deep-copy device caches
    try:
        cp = deepcopy(model).cpu()
    finally:
        for m in model.modules():
            m.__dict__.update(live[id(m)])
    cp.load_state_dict({k: v.detach().cpu().float() if v.is_floating_point() else v.detach().cpu() for k, v in state_dict.items()})
    for m in cp.modules():
        for k in _DROP_ATTRS:
            m.__dict__.pop(k, None)
        if isinstance(m, PlainConv2d):
            m.__class__ = nn.Conv2d
        elif isinstance(m, Upsample):
            m.__class__ = nn.Upsample
            m.__dict__.setdefault("name", "Upsample")
            m.scale_factor = float(m.scale_factor)
            m.__dict__.setdefault("align_corners", None)
            m.__dict__.setdefault("recompute_scale_factor", None)
        t = getattr(m, "type", None)
        if isinstance(t, str) and "." not in t:
            m.type = "torch.nn.modules.upsampling.Upsample" if t == "nn.Upsample" else f"{_REF_PATHS.get(t, 'ultralytics.nn.modules')}.{t}"
    cp.eval()
    for p in cp.parameters():
        p.requires_grad_(False)
    return cp.half() if half else cp


def save_reference_checkpoint(path, model: nn.Module, state_dict: Dict[str, torch.Tensor], extra: Dict[str, Any] = None) -> None:
    """Write ``path`` with the keys of ``BaseTrainer.save_model`` (engine/trainer.py:514-545): ``model`` None, ``ema`` the module
    graph in fp16 — pickled under the reference's class paths, so the reference's own ``torch.load`` / ``attempt_load_one_weight``
    (nn/tasks.py:786-926) restore it as ITS classes — optimizer state in fp16 (torch_utils.py:553-566), train_args as a dict."""
    ck = {"epoch": -1, "best_fitness": None, "model": None, "ema": reference_module_graph(model, state_dict) if model is not None else None, "updates": 0,
          "optimizer": None,
          "train_args": {}, "train_metrics": {}, "train_results": {}, "date": None, "version": "8.3.0", "license": "AGPL-3.0 (https://ultralytics.com/license)",
          "docs": "https://docs.ultralytics.com"}
    ck.update(extra or {})
    opt = ck.get("optimizer")
    if isinstance(opt, dict):  # convert_optimizer_state_dict_to_fp16: every fp32 state tensor except 'step'
        for st in opt.get("state", {}).values():
            for k, v in st.items():
                if k != "step" and isinstance(v, torch.Tensor) and v.dtype == torch.float32:
                    st[k] = v.detach().cpu().half()
                elif isinstance(v, torch.Tensor):
                    st[k] = v.detach().cpu()
    ck["train_args"] = {k: (str(v) if isinstance(v, (torch.dtype, torch.device)) else v) for k, v in dict(ck.get("train_args") or {}).items()}
    with _reference_class_paths():
        torch.save(ck, str(path))
