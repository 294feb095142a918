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
