"""Training loop of the Drone-YOLO path on the device (reference: ultralytics/engine/trainer.py ``BaseTrainer`` and
models/yolo/detect/train.py ``DetectionTrainer``).

What is kept from the reference, with the place it comes from:
  * one process per GPU, batch // world_size images per rank (trainer.py:286), loss * world_size followed by DDP's
    mean == a SUM all-reduce of the per-rank gradients (trainer.py:382-383 + DDP) — here ONE flat fp32 gradient buffer
    is all-reduced over RCCL/xGMI after backward (43 MB for Drone-YOLO-s; no SyncBN, as in the reference);
  * build_optimizer (trainer.py:764-825): three groups — biases (no decay), BatchNorm weights (no decay), other weights
    (decay); optimizer 'auto' = SGD(lr0, momentum, nesterov) beyond 10,000 iterations else AdamW(0.002*5/(4+nc), betas
    (momentum, 0.999)); weight_decay scaled by batch*accumulate/nbs (trainer.py:254-256);
  * warm-up of lr / momentum per iteration and the linear lr schedule (trainer.py:361-372, 214-216);
  * gradient clipping max_norm 10 (trainer.py:594), ModelEMA (utils/torch_utils.py:515-545) on rank 0's replica (every
    rank keeps it here: replicas are identical, so no broadcast is needed).
Parameters, gradients, BatchNorm buffers and the EMA copy live in FLAT fp32 buffers (the module parameters are views), so
the clip norm is one reduction, each optimizer group one kernel launch, the EMA one launch and the all-reduce one call.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import hip_ops as H
from .. import parallel as P


def get_cfg(overrides: Optional[dict] = None) -> dict:
    """cfg/default.yaml (the reference's key names and values, ultralytics/cfg/default.yaml) updated with overrides."""
    import os

    import yaml

    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfg", "default.yaml")) as f:
        cfg = yaml.safe_load(f)
    unknown = set(overrides or {}) - set(cfg)
    if unknown:
        raise KeyError(f"unknown training argument(s) {sorted(unknown)}")
    cfg.update(overrides or {})
    return cfg


def param_group_names(model: nn.Module) -> Tuple[List[str], List[str], List[str]]:
    """(decay weights, norm weights, biases) in module order — build_optimizer's split (trainer.py:795-808)."""
    g0, g1, g2 = [], [], []
    bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for mname, m in model.named_modules():
        for pname, p in m.named_parameters(recurse=False):
            if not p.requires_grad:
                continue
            full = f"{mname}.{pname}" if mname else pname
            if "bias" in full:
                g2.append(full)
            elif isinstance(m, bn):
                g1.append(full)
            else:
                g0.append(full)
    return g0, g1, g2


class FlatState:
    """Parameters / gradients (and BatchNorm buffers) of a model re-homed into flat fp32 device buffers."""

    def __init__(self, model: nn.Module, device):
        self.groups = param_group_names(model)
        params = dict(model.named_parameters())
        order = [k for g in self.groups for k in g]
        self.sizes = [sum(params[k].numel() for k in g) for g in self.groups]
        n = sum(self.sizes)
        self.P = torch.empty(n, dtype=torch.float32, device=device)
        self.G = torch.zeros(n, dtype=torch.float32, device=device)
        off = 0
        for k in order:
            p = params[k]
            c = p.numel()
            self.P[off : off + c].copy_(p.detach().reshape(-1))
            p.data = self.P[off : off + c].view_as(p)
            p.grad = self.G[off : off + c].view_as(p)
            off += c
        bufs = [(k, b) for k, b in model.named_buffers() if b.is_floating_point()]
        nb = sum(b.numel() for _, b in bufs)
        self.B = torch.empty(max(nb, 1), dtype=torch.float32, device=device)
        off = 0
        for _, b in bufs:
            c = b.numel()
            self.B[off : off + c].copy_(b.detach().reshape(-1).float())
            b.data = self.B[off : off + c].view_as(b)
            off += c
        self.nb = nb

    def group_slices(self):
        off = 0
        for s in self.sizes:
            yield slice(off, off + s)
            off += s


class ModelEMA:
    """Exponential moving average of parameters and floating buffers — torch_utils.py:515-545 (decay 0.9999, tau 2000)."""

    def __init__(self, flat: FlatState, decay: float = 0.9999, tau: float = 2000.0, updates: int = 0):
        self.P, self.B = flat.P.clone(), flat.B.clone()
        self.flat, self.decay, self.tau, self.updates = flat, decay, tau, updates

    def update(self) -> None:
        self.updates += 1
        d = self.decay * (1 - math.exp(-self.updates / self.tau))
        H.ema_update_(self.P, self.flat.P, d)
        if self.flat.nb:
            H.ema_update_(self.B, self.flat.B, d)


class DetectionTrainer:
    """Minimal trainer for tensor batches: ``step(batch)`` = forward, loss, backward, all-reduce, clip, optimizer, EMA."""

    def __init__(self, model: nn.Module, overrides: Optional[dict] = None, iterations_hint: int = 0):
        self.args = a = get_cfg(overrides or {})
        self.rank, self.local_rank, self.world = P.dist_env()
        self.device = torch.device("cuda", self.local_rank if a.get("device", "") in ("", None) else int(str(a["device"]).split(",")[0]))
        self.model = model.to(self.device).train()
        self.model.train_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.get("dtype", "bf16")] if a.get("amp", True) \
            else torch.float32
        self.model.args = type("Args", (), dict(box=a["box"], cls=a["cls"], dfl=a["dfl"]))()
        self.batch_size = int(a["batch"])  # GLOBAL batch, as in the reference; each rank sees batch // world
        self.accumulate = max(round(a["nbs"] / max(self.batch_size, 1)), 1)
        self.weight_decay = a["weight_decay"] * self.batch_size * self.accumulate / a["nbs"]
        self.flat = FlatState(self.model, self.device)
        nc = self.model.yaml["nc"]
        name = a.get("optimizer", "auto")
        if name == "auto":  # trainer.py:784-793
            if iterations_hint > 10000:
                name, self.lr0, self.momentum = "SGD", 0.01, 0.9
            else:
                name, self.lr0, self.momentum = "AdamW", round(0.002 * 5 / (4 + nc), 6), 0.9
        else:
            self.lr0, self.momentum = a["lr0"], a["momentum"]
        if name not in ("SGD", "AdamW"):
            raise NotImplementedError(f"optimizer {name}: SGD and AdamW are built")
        self.opt_name = name
        n = self.flat.P.numel()
        self.buf1 = torch.zeros(n, dtype=torch.float32, device=self.device)  # momentum / first moment
        self.buf2 = torch.zeros(n, dtype=torch.float32, device=self.device) if name == "AdamW" else None
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.ema = ModelEMA(self.flat)
        self.opt_steps = 0
        self.iters = 0
        self.epochs = int(a["epochs"])
        self.lf = lambda x: max(1 - x / self.epochs, 0) * (1.0 - a["lrf"]) + a["lrf"]  # linear schedule, trainer.py:214-216

    # ---- schedules -----------------------------------------------------------------------------------------------------
    def lr_momentum(self, ni: int, epoch: int, nb: int) -> Tuple[List[float], float]:
        """Per-group lr (decay weights, norm weights, biases) and momentum at iteration ni — trainer.py:361-372."""
        a = self.args
        nw = max(round(a["warmup_epochs"] * nb), 100) if a["warmup_epochs"] > 0 else -1
        target = self.lr0 * self.lf(epoch)
        if ni <= nw:
            xi = [0, nw]
            lrs = [float(np.interp(ni, xi, [0.0, target])), float(np.interp(ni, xi, [0.0, target])),
                   float(np.interp(ni, xi, [a["warmup_bias_lr"], target]))]
            mom = float(np.interp(ni, xi, [a["warmup_momentum"], self.momentum]))
            return lrs, mom
        return [target] * 3, self.momentum

    # ---- one iteration ---------------------------------------------------------------------------------------------------
    def step(self, batch: Dict[str, torch.Tensor], epoch: int = 0, nb: int = 1000):
        """batch: img (N_local, 3, H, W) uint8/float on the device, batch_idx / cls / bboxes as the reference's collate gives
        them (data/dataset.py:232-248).  Returns (loss, loss_items) of this rank."""
        if self.iters % self.accumulate == 0:
            self.flat.G.zero_()
        loss, items = self.model(batch)
        # the reference multiplies by world_size and lets DDP average: the net effect is the plain SUM below
        loss.backward()
        self.iters += 1
        if self.iters % self.accumulate == 0:
            self.optimizer_step(epoch, nb)
        return loss.detach(), items

    def optimizer_step(self, epoch: int = 0, nb: int = 1000) -> None:
        G, Pm = self.flat.G, self.flat.P
        P.allreduce_gradients(G)  # RCCL ring over xGMI: one 4*n_params-byte bucket (no-op in a single process)
        self.sumsq.zero_()
        H.sumsq_into(self.sumsq, G)
        lrs, mom = self.lr_momentum(self.iters // self.accumulate - 1, epoch, nb)
        self.opt_steps += 1
        for gi, sl in enumerate(self.flat.group_slices()):
            if sl.stop == sl.start:
                continue
            wd = self.weight_decay if gi == 0 else 0.0
            if self.opt_name == "SGD":
                H.sgd_step_(Pm[sl], G[sl], self.buf1[sl], lrs[gi], mom, wd, True, self.opt_steps == 1, self.sumsq, 10.0)
            else:
                H.adamw_step_(Pm[sl], G[sl], self.buf1[sl], self.buf2[sl], lrs[gi], (mom, 0.999), 1e-8, wd, self.opt_steps, self.sumsq, 10.0)
        self.ema.update()
