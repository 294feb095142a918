"""Device / weight helpers (reference: ultralytics/utils/torch_utils.py)."""
from __future__ import annotations

import os
import random

import numpy as np
import torch
import torch.nn as nn


def select_device(device="", batch=0, newline=False, verbose=True) -> torch.device:
    """'' | 0 | '0' | 'cuda:0' | torch.device -> torch.device — reference torch_utils.py:133-239.

    The accelerated path is HIP only: 'cpu' (or no visible GPU) raises instead of silently running
    a slow fallback.  Like the reference (torch_utils.py:202-219) a multi-GPU string such as '0,1'
    selects the first index for single-process inference; multi-GPU inference is one process per
    GPU (see parallel.py), training shards through torch.distributed.
    """
    if isinstance(device, torch.device):
        dev = device
    else:
        s = str(device).lower().replace("cuda:", "").replace("(", "").replace(")", "").replace(" ", "")
        if s in ("cpu", "mps"):
            raise RuntimeError(f"device='{s}' requested: this package accelerates the Drone-YOLO path on MI355X (HIP) "
                               "only and has no CPU fallback")
        idx = int(s.split(",")[0]) if s not in ("", "none") else int(os.environ.get("LOCAL_RANK", 0) or 0)
        dev = torch.device("cuda", max(idx, 0))
    if dev.type != "cuda":
        raise RuntimeError(f"device '{dev}' is not a HIP device; no CPU fallback exists for this path")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible (torch.cuda.is_available() is False); this path needs an MI355X")
    return dev


def fuse_conv_and_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """Conv2d + BatchNorm2d -> one Conv2d with bias — reference torch_utils.py:242-269."""
    from ..nn.modules.conv import fold_conv_bn

    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                      dilation=conv.dilation, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    w, b = fold_conv_bn(conv.weight, conv.bias, bn)
    fused.weight.copy_(w)
    fused.bias.copy_(b)
    return fused


def initialize_weights(model: nn.Module) -> None:
    """BatchNorm eps=1e-3, momentum=0.03; in-place activations — reference torch_utils.py:423-433."""
    for m in model.modules():
        t = type(m)
        if t is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif t in {nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU}:
            m.inplace = True


def init_seeds(seed=0, deterministic=False) -> None:
    """Seed python / numpy / torch — reference torch_utils.py:487-512."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def strip_optimizer(f="best.pt", s: str = "", updates: dict = None) -> dict:
    """Finalise a training checkpoint — reference torch_utils.py:553-616: ``model`` := the EMA module graph in fp16 with frozen
    parameters, ``optimizer`` / ``best_fitness`` / ``ema`` / ``updates`` := None, ``epoch`` := -1, ``train_args`` merged over the defaults,
    fresh ``date`` / ``version`` metadata, ``updates`` overlaid last; written back to ``s or f`` under the reference's class paths (so the
    reference's own ``torch.load`` restores its classes).  This package's exact-resume state (``dyolo_state``, engine/trainer.py) is
    dropped with the optimizer.  Returns the combined dict ({} for a file that is no checkpoint, with a warning, as the reference)."""
    from datetime import datetime

    from ..nn import checkpoint as CK
    from . import LOGGER

    try:
        x = CK.read_checkpoint_dict(str(f))
        assert "model" in x, "'model' missing from checkpoint"
    except Exception as e:  # noqa: BLE001 (the reference skips anything it cannot read)
        LOGGER.warning(f"WARNING Skipping {f}, not a valid checkpoint: {e}")
        return {}
    metadata = {"date": datetime.now().isoformat(), "version": "8.3.0", "license": "AGPL-3.0 License (https://ultralytics.com/license)",
                "docs": "https://docs.ultralytics.com"}
    if x.get("ema"):
        x["model"] = x["ema"]  # replace model with EMA
    model = x["model"]
    if not isinstance(model, nn.Module):
        LOGGER.warning(f"WARNING Skipping {f}: no module graph under 'ema' / 'model'")
        return {}
    if hasattr(model, "args") and not isinstance(model.args, dict):
        model.args = dict(getattr(model.args, "__dict__", {}))  # (IterableSimpleNamespace -> dict)
    if hasattr(model, "criterion"):
        model.criterion = None  # strip loss criterion
    model.half()
    for p in model.parameters():
        p.requires_grad = False
    from ..engine.trainer import get_cfg

    args = {**get_cfg({}), **{k: v for k, v in dict(x.get("train_args") or {}).items()}}
    for k in ("optimizer", "best_fitness", "ema", "updates"):
        x[k] = None
    x.pop("dyolo_state", None)
    x["epoch"] = -1
    x["train_args"] = args
    combined = {**metadata, **x, **(updates or {})}
    with CK._reference_class_paths():
        torch.save(combined, str(s or f))
    mb = os.path.getsize(str(s or f)) / 1e6
    LOGGER.info(f"Optimizer stripped from {f},{f' saved as {s},' if s else ''} {mb:.1f}MB")
    return combined


class EarlyStopping:
    """Stop when ``patience`` epochs have passed without a fitness improvement — reference torch_utils.py:733-776."""

    def __init__(self, patience=50):
        self.best_fitness = 0.0  # i.e. mAP
        self.best_epoch = 0
        self.patience = patience or float("inf")  # epochs to wait after fitness stops improving to stop
        self.possible_stop = False  # possible stop may occur next epoch

    def __call__(self, epoch, fitness) -> bool:
        if fitness is None:  # val=False
            return False
        if fitness > self.best_fitness or self.best_fitness == 0:  # allow for early zero-fitness stage of training
            self.best_epoch = epoch
            self.best_fitness = fitness
        delta = epoch - self.best_epoch  # epochs without improvement
        self.possible_stop = delta >= (self.patience - 1)
        stop = delta >= self.patience
        if stop:
            from . import LOGGER

            LOGGER.info(f"EarlyStopping: Training stopped early as no improvement observed in last {self.patience} epochs. "
                        f"Best results observed at epoch {self.best_epoch}, best model saved as best.pt.")
        return stop
