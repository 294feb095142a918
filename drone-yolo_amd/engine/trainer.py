"""Training loop of the Drone-YOLO path on the device (reference: ultralytics/engine/trainer.py ``BaseTrainer`` and
models/yolo/detect/train.py ``DetectionTrainer``).

What is kept from the reference, with the place it comes from:
  * ``train()`` (trainer.py:171-207): ``device="0,1,.."`` with no LOCAL_RANK in the environment makes this process the
    launcher — it writes a temp script that rebuilds the trainer from its arguments and runs it under
    ``torch.distributed.run`` with one rank per GPU (utils/dist.py:25-66), as a CHILD process; otherwise ``_do_train``;
  * ``_do_train`` (trainer.py:319-476): epochs x batches, warm-up by batch counter ``ni = i + nb * epoch`` (accumulate ramped
    from 1 to nbs / batch, bias lr from ``warmup_bias_lr``, other lrs from 0, SGD momentum from ``warmup_momentum`` to
    ``momentum``), optimizer step every ``accumulate`` batches, linear lr schedule per epoch (trainer.py:214-216),
    ``results.csv`` (trainer.py:700-708), ``last.pt`` (trainer.py:514-545);
  * one process per GPU, batch // world_size images per rank (trainer.py:286), loss * world_size followed by DDP's mean == a
    SUM all-reduce of the per-rank gradients (trainer.py:382-383 + DDP): the flat fp32 gradient buffer is all-reduced over
    RCCL/xGMI in buckets (reverse layer order), each bucket issued as soon as backward has produced its last gradient, so
    the ring runs under the remaining backward kernels (parallel.GradBuckets); no SyncBN, as in the reference;
  * build_optimizer (trainer.py:764-825): three groups — biases (no decay), BatchNorm weights (no decay), other weights
    (decay); optimizer 'auto' = SGD(0.01, 0.9, nesterov) beyond 10,000 iterations else AdamW(0.002*5/(4+nc), betas
    (0.9, 0.999)) with ``warmup_bias_lr`` forced to 0; weight_decay scaled by batch*accumulate/nbs (trainer.py:254-256);
  * gradient clipping max_norm 10 (trainer.py:594), ModelEMA (utils/torch_utils.py:515-545).
Parameters, gradients, BatchNorm buffers and the EMA copy live in FLAT fp32 buffers (the module parameters are views), so
the clip norm is one reduction, each optimizer group one kernel launch and the EMA one launch.
  * validation inside the loop (trainer.py:427-442, 605-615; r04): rank 0 evaluates the EMA weights every epoch (``val: true``) with the
    validator's NMS (multi_label, conf 0.001) — mAP50 / mAP50-95 / fitness into results.csv, ``best.pt`` beside ``last.pt``
    (engine/validator.py).
Out of scope: dataset files, augmentation, callbacks, plots, early stopping.  The loader here serves
tensor datasets (uint8 images + labels in the reference's collate layout) or the synthetic VisDrone-shaped set of SURVEY §8(d).
"""
from __future__ import annotations

import csv
import math
import os
import subprocess
import time
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import parallel as P
from ..utils import LOGGER


def get_cfg(overrides: Optional[dict] = None) -> dict:
    """cfg/default.yaml (the reference's key names and values, ultralytics/cfg/default.yaml) updated with overrides."""
    import yaml

    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfg", "default.yaml")) as f:
        cfg = yaml.safe_load(f)
    unknown = set(overrides or {}) - set(cfg)
    if unknown:
        raise KeyError(f"unknown training argument(s) {sorted(unknown)}")
    cfg.update(overrides or {})
    check_cfg(cfg)
    return cfg


# argument classes of the reference's check_cfg (ultralytics/cfg/__init__.py:147-236, 324-395), for the keys this path reads
CFG_FLOAT_KEYS = frozenset({"warmup_epochs", "box", "cls", "dfl", "time", "batch"})  # int or float
CFG_FRACTION_KEYS = frozenset({"lr0", "lrf", "momentum", "weight_decay", "warmup_momentum", "warmup_bias_lr", "conf", "iou"})  # 0.0 <= v <= 1.0
CFG_INT_KEYS = frozenset({"epochs", "patience", "seed", "max_det", "nbs"})
CFG_BOOL_KEYS = frozenset({"save", "verbose", "single_cls", "half", "agnostic_nms", "stream", "amp", "multi_scale"})


def check_cfg(cfg: dict, hard: bool = True) -> None:
    """Type and range checks of the arguments, as the reference's ``check_cfg`` (cfg/__init__.py:324-395): None is an unset optional; a float key takes int or
    float, a fraction key additionally 0 <= v <= 1, an int key an int, a bool key a bool.  ``hard`` raises; otherwise the value is converted in place."""
    for k, v in cfg.items():
        if v is None:
            continue
        if k in CFG_FLOAT_KEYS or k in CFG_FRACTION_KEYS:
            if isinstance(v, bool) or not isinstance(v, (int, float)):
                if hard:
                    raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}. Valid '{k}' types are int (i.e. '{k}=0') or float (i.e. '{k}=0.5')")
                cfg[k] = v = float(v)
            if k in CFG_FRACTION_KEYS and not (0.0 <= v <= 1.0):
                raise ValueError(f"'{k}={v}' is an invalid value. Valid '{k}' values are between 0.0 and 1.0.")
        elif k in CFG_INT_KEYS and (isinstance(v, bool) or not isinstance(v, int)):
            if hard:
                raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}. '{k}' must be an int (i.e. '{k}=8')")
            cfg[k] = int(v)
        elif k in CFG_BOOL_KEYS and not isinstance(v, bool):
            if hard:
                raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}. '{k}' must be a bool (i.e. '{k}=True' or '{k}=False')")
            cfg[k] = bool(v)


def param_group_names(model: nn.Module, include_frozen: bool = False) -> Tuple[List[str], List[str], List[str]]:
    """(decay weights, norm weights, biases) in module order — build_optimizer's split (trainer.py:795-808).  The reference does
    not filter on requires_grad (its decay group also holds the frozen ``model.N.dfl.conv.weight``): ``include_frozen`` gives that
    enumeration (the optimizer state-dict's indices); the flat training buffers hold the trainable parameters only."""
    g0, g1, g2 = [], [], []
    bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for mname, m in model.named_modules():
        for pname, p in m.named_parameters(recurse=False):
            if not p.requires_grad and not include_frozen:
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
        self.offsets: Dict[str, Tuple[int, int]] = {}  # name -> (offset, numel) in P / G
        self.reg_order = [k for k in params if params[k].requires_grad]  # registration (= forward execution) order
        self.params: Dict[str, nn.Parameter] = {}
        off = 0
        for k in order:
            p = params[k]
            c = p.numel()
            self.P[off : off + c].copy_(p.detach().reshape(-1))
            p.data = self.P[off : off + c].view_as(p)
            p.grad = self.G[off : off + c].view_as(p)
            self.offsets[k] = (off, c)
            self.params[k] = p
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

    def enable_sink(self) -> None:
        """A second flat buffer the backward kernels write this batch's parameter gradients into (nn/autograd_ops.py::grad_sink:
        weight gradients in the wgrad kernels' (cout, k, k, cin) order, BatchNorm gradients as they are), folded into G by ONE
        ``dy_grad_sink_flush`` launch per batch (``flush_sink``) — instead of autograd's AccumulateGrad per parameter: ~240
        element-wise launches plus the zero-fills of as many temporaries per step.  With several ranks ``parallel.GradBuckets`` flushes
        the sink bucket by bucket (``use_sink``) and all-reduces each bucket behind its flush."""
        self.S = torch.zeros_like(self.G)
        rows = []
        for k, (off, c) in self.offsets.items():
            p = self.params[k]
            p._dy_sink = self.S[off : off + c]
            if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
                rows.append((off, p.shape[0], p.shape[1], p.shape[2] * p.shape[3]))
            else:
                rows.append((off, c, 1, 1))
        self.sink_entries = torch.tensor(rows, dtype=torch.int64, device=self.G.device)

    def flush_sink(self) -> None:
        if getattr(self, "S", None) is not None:
            from .. import hip_ops as H

            H.join_side_stream()  # the weight-gradient kernels run on a second stream (hip_ops.conv_wgrad_into)
            H.grad_sink_flush_(self.sink_entries, self.G, self.S)


class ModelEMA:
    """Exponential moving average of parameters and floating buffers — torch_utils.py:515-545 (decay 0.9999, tau 2000)."""

    def __init__(self, flat: FlatState, decay: float = 0.9999, tau: float = 2000.0, updates: int = 0):
        self.P, self.B = flat.P.clone(), flat.B.clone()
        self.flat, self.decay, self.tau, self.updates = flat, decay, tau, updates

    def update(self) -> None:
        from .. import hip_ops as H

        self.updates += 1
        d = self.decay * (1 - math.exp(-self.updates / self.tau))
        H.ema_update_(self.P, self.flat.P, d)
        if self.flat.nb:
            H.ema_update_(self.B, self.flat.B, d)

    def state_dict(self, model: nn.Module) -> Dict[str, torch.Tensor]:
        """The EMA weights under the model's state-dict keys (what ``deepcopy(self.ema.ema)`` holds in the reference)."""
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for k, (off, c) in self.flat.offsets.items():
            sd[k] = self.P[off : off + c].view_as(sd[k]).clone()
        off = 0
        for k, b in model.named_buffers():
            if b.is_floating_point():
                c = b.numel()
                sd[k] = self.B[off : off + c].view_as(b).clone()
                off += c
        return sd


# ---- data ----------------------------------------------------------------------------------------------------------------
def synthetic_dataset(n: int, imgsz: int, seed: int, nc: int = 10) -> Dict[str, torch.Tensor]:
    """uint8 images randint(0, 256); per image n ~ Poisson(50) clipped to [1, 300] boxes (VisDrone-like
    density), class uniform, centres uniform(0.05, 0.95), wh lognormal(median 0.03, sigma 0.6) clipped to [0.004, 0.5]."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (n, 3, imgsz, imgsz), generator=g, dtype=torch.uint8)
    counts = torch.poisson(torch.full((n,), 50.0), generator=g).clamp(1, 300).long()
    m = int(counts.sum())
    bi = torch.repeat_interleave(torch.arange(n), counts).float()
    cls = torch.randint(0, nc, (m, 1), generator=g).float()
    cxy = torch.rand(m, 2, generator=g) * 0.9 + 0.05
    wh = (torch.randn(m, 2, generator=g) * 0.6 + math.log(0.03)).exp().clamp(0.004, 0.5)
    return dict(img=img, batch_idx=bi, cls=cls, bboxes=torch.cat((cxy, wh), 1))


def load_dataset(data, imgsz: int, nc: int, seed: int) -> Dict[str, torch.Tensor]:
    """``data``: a dict / a ``.pt`` file of tensors ``img`` (N,3,H,W) uint8, ``batch_idx`` (M,), ``cls`` (M,1), ``bboxes`` (M,4
    normalised xywh) — the reference's collate layout (data/dataset.py:232-248) over the whole set — or ``"synthetic[:N]"``."""
    if isinstance(data, dict):
        d = data
    elif isinstance(data, str) and data.startswith("synthetic"):
        n = int(data.split(":")[1]) if ":" in data else 128
        d = synthetic_dataset(n, imgsz, seed=1000 + seed, nc=nc)
    elif isinstance(data, (str, Path)) and str(data).endswith(".pt"):
        d = torch.load(str(data), map_location="cpu", weights_only=True)
    else:
        raise NotImplementedError(f"data={data!r}: tensor datasets (.pt / dict) and 'synthetic[:N]' are built; image folders and dataset YAMLs "
                                  "(decoding, mosaic, augmentation) are outside the accelerated path")
    missing = {"img", "batch_idx", "cls", "bboxes"} - set(d)
    if missing:
        raise KeyError(f"dataset lacks {sorted(missing)}")
    if d["img"].dtype != torch.uint8 or d["img"].dim() != 4:
        raise ValueError("dataset 'img' must be uint8 (N, 3, H, W)")
    return d


class TensorLoader:
    """Batches of a tensor dataset with ``DistributedSampler`` semantics (data/build.py:144: shuffle by seed + epoch, pad to a
    multiple of world size by wrapping around, rank r takes indices r::world); labels re-indexed per batch as the reference's
    collate_fn does (``batch_idx`` = position in the batch)."""

    def __init__(self, data: Dict[str, torch.Tensor], batch: int, rank: int = 0, world: int = 1, seed: int = 0, shuffle: bool = True):
        self.d, self.batch, self.rank, self.world, self.seed, self.shuffle, self.epoch = data, max(int(batch), 1), rank, world, seed, shuffle, 0
        self.n = data["img"].shape[0]
        self.per_rank = -(-self.n // world)
        bi = data["batch_idx"].long()
        order = torch.argsort(bi, stable=True)
        self._rows = order
        counts = torch.bincount(bi, minlength=self.n)
        self._start = torch.cat((torch.zeros(1, dtype=torch.long), counts.cumsum(0)))

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return -(-self.per_rank // self.batch)

    def indices(self) -> List[int]:
        if self.shuffle:
            idx = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            idx = list(range(self.n))
        total = self.per_rank * self.world
        idx += idx[: total - len(idx)]
        return idx[self.rank : total : self.world]

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        idx = self.indices()
        for s in range(0, len(idx), self.batch):
            take = idx[s : s + self.batch]
            rows, bi = [], []
            for j, i in enumerate(take):
                r = self._rows[self._start[i] : self._start[i + 1]]
                rows.append(r)
                bi.append(torch.full((len(r),), float(j)))
            rows = torch.cat(rows) if rows else torch.zeros(0, dtype=torch.long)
            yield dict(img=self.d["img"][take], batch_idx=torch.cat(bi) if bi else torch.zeros(0), cls=self.d["cls"][rows].view(-1, 1).float(),
                       bboxes=self.d["bboxes"][rows].float())


# ---- optimizer selection and schedules (host logic only) -------------------------------------------------------------------
def resolve_optimizer(args: dict, nc: int, iterations: float) -> Tuple[str, float, float, float]:
    """(name, lr0, momentum, warmup_bias_lr) — build_optimizer's 'auto' rule (trainer.py:784-793): SGD(0.01, 0.9) beyond 10,000
    iterations, else AdamW(round(0.002 * 5 / (4 + nc), 6), 0.9); either way warmup_bias_lr is forced to 0."""
    name = args.get("optimizer", "auto")
    if name == "auto":
        name, lr0, mom = ("SGD", 0.01, 0.9) if iterations > 10000 else ("AdamW", round(0.002 * 5 / (4 + nc), 6), 0.9)
        return name, lr0, mom, 0.0
    return name, args["lr0"], args["momentum"], args["warmup_bias_lr"]


class OptimSchedule:
    """What the reference keeps in ``optimizer.param_groups`` between iterations and how its loop mutates it: LambdaLR at the
    top of an epoch (trainer.py:349, lr = lr0 * lf(epoch) for every group) and the per-batch warm-up (trainer.py:362-377).
    Group order here is (decay weights, norm weights, biases); the reference's is (biases, decay weights, norm weights)."""

    def __init__(self, args: dict, opt_name: str, lr0: float, momentum: float, warmup_bias_lr: float, batch_size: int, epochs: int):
        self.args, self.opt_name, self.lr0, self.momentum, self.warmup_bias_lr = args, opt_name, lr0, momentum, warmup_bias_lr
        self.batch_size, self.epochs = batch_size, epochs
        self.accumulate = max(round(args["nbs"] / max(batch_size, 1)), 1)  # trainer.py:254
        self.cur_lrs = [lr0] * 3
        self.cur_momentum = momentum  # SGD's; AdamW groups have no 'momentum' key, so its beta1 is never warmed up
        lrf = args["lrf"]
        self.lf = lambda x: max(1 - x / self.epochs, 0) * (1.0 - lrf) + lrf  # linear, trainer.py:214-216

    def warmup_iters(self, nb: int) -> int:
        return max(round(self.args["warmup_epochs"] * nb), 100) if self.args["warmup_epochs"] > 0 else -1

    def scheduler_step(self, epoch: int) -> None:
        self.cur_lrs = [self.lr0 * self.lf(epoch)] * 3

    def warmup(self, ni: int, epoch: int, nb: int) -> None:
        nw = self.warmup_iters(nb)
        if ni > nw:
            return
        xi, a = [0, nw], self.args
        self.accumulate = max(1, int(np.interp(ni, xi, [1, a["nbs"] / self.batch_size]).round()))
        target = self.lr0 * self.lf(epoch)
        self.cur_lrs = [float(np.interp(ni, xi, [0.0, target])), float(np.interp(ni, xi, [0.0, target])),
                        float(np.interp(ni, xi, [self.warmup_bias_lr, target]))]
        if self.opt_name == "SGD":
            self.cur_momentum = float(np.interp(ni, xi, [a["warmup_momentum"], a["momentum"]]))


# ---- trainer -------------------------------------------------------------------------------------------------------------
class DetectionTrainer:
    """``DetectionTrainer(model, overrides)`` for tensor batches (``step``) or ``DetectionTrainer(overrides=...)`` +
    ``train()`` for the whole loop (model from ``overrides['model']``, data from ``overrides['data']``)."""

    def __init__(self, model: Optional[nn.Module] = None, overrides: Optional[dict] = None, iterations_hint: int = 0):
        self.args = get_cfg(overrides or {})
        self._resume_ckpt = None
        self.check_resume(overrides or {})  # may replace self.args by the checkpoint's (trainer.py:105)
        a = self.args
        data = a.get("data")
        if isinstance(data, (str, Path)) and not (str(data).startswith("synthetic") or str(data).endswith(".pt")):
            load_dataset(data, 0, 0, 0)  # raises NotImplementedError naming what is built, before any device work
        if not torch.cuda.is_available():
            raise RuntimeError("training needs an MI355X: no HIP device visible and this path has no CPU fallback")
        self.rank, self.local_rank, self.world = P.dist_env()
        self.model = model
        self.iterations_hint = iterations_hint
        self.save_dir = Path(a.get("project") or "runs/detect") / (a.get("name") or "train")
        self.wdir = self.save_dir / "weights"
        self.last, self.best, self.csv = self.wdir / "last.pt", self.wdir / "best.pt", self.save_dir / "results.csv"
        self.best_fitness, self.fitness, self.metrics = None, None, {}
        self.validator = None
        self.val_loader = None
        self.epochs = int(a["epochs"])
        self.start_epoch, self.epoch = 0, 0
        self.loss_names = ("box_loss", "cls_loss", "dfl_loss")
        from ..utils.torch_utils import EarlyStopping

        self.stopper, self.stop = EarlyStopping(patience=a.get("patience", 100)), False  # trainer.py:314
        self.flat = None
        self.tloss = None
        self.train_loader = None
        if model is not None:
            self._setup_model_state()

    # ---- resume ----------------------------------------------------------------------------------------------------------
    def check_resume(self, overrides: dict) -> None:
        """trainer.py:697-729: ``resume`` = True (with ``model`` = .../last.pt) or the path of a last.pt.  The checkpoint's ``train_args``
        become the arguments, ``model`` / ``resume`` point at the file, and imgsz / batch / device may be overridden (less memory, another
        GPU).  A tensor dataset handed over as a dict is not in the file: ``data`` from the overrides is kept when the checkpoint has none."""
        resume = self.args.get("resume")
        if resume:
            try:
                cand = resume if isinstance(resume, (str, Path)) and Path(str(resume)).exists() else self.args.get("model")
                last = Path(str(cand))
                if not (last.suffix == ".pt" and last.exists()):
                    raise FileNotFoundError(str(cand))
                from ..nn.checkpoint import read_checkpoint_dict

                ckpt = read_checkpoint_dict(str(last))
                ck_args = {k: v for k, v in dict(ckpt.get("train_args") or {}).items() if k in self.args}  # (keys this path knows)
                if isinstance(overrides.get("data"), dict) or not ck_args.get("data"):
                    ck_args["data"] = overrides.get("data", self.args.get("data"))
                self.args = get_cfg(ck_args)
                self.args["model"] = self.args["resume"] = str(last)
                for k in ("imgsz", "batch", "device"):
                    if k in overrides:
                        self.args[k] = overrides[k]
                self._resume_ckpt = ckpt
                resume = True
            except Exception as e:
                raise FileNotFoundError("Resume checkpoint not found. Please pass a valid checkpoint to resume from, i.e. "
                                        "YOLO('path/to/last.pt').train(resume=True)") from e
        self.resume = bool(resume)

    def resume_training(self, ckpt: Optional[dict]) -> None:
        """trainer.py:731-754: optimizer state, EMA weights + ``updates``, ``best_fitness`` and the next epoch from a last.pt.  A file written
        by this trainer also carries ``dyolo_state`` — the fp32 live weights, BatchNorm buffers, EMA and optimizer moments, the GradScaler
        state and the step counters — and then the run continues as if it had never stopped; a file written by the reference has the fp16
        EMA graph and fp16 optimizer state only, and resumes as the reference does (the live weights restart from the EMA, tasks.py:906)."""
        if ckpt is None or not self.resume:
            return
        best_fitness = 0.0
        start_epoch = int(ckpt.get("epoch", -1)) + 1
        exact = ckpt.get("dyolo_state")
        dev = self.device
        if ckpt.get("optimizer") is not None:
            self.load_optimizer_state_dict(ckpt["optimizer"])
            best_fitness = ckpt.get("best_fitness")
        if ckpt.get("ema") is not None:
            sd = {k: v.float() for k, v in ckpt["ema"].state_dict().items() if v.is_floating_point()}
            for k, (off, c) in self.flat.offsets.items():
                self.ema.P[off : off + c].copy_(sd[k].reshape(-1))
            off = 0
            for k, b in self.model.named_buffers():
                if b.is_floating_point():
                    c = b.numel()
                    self.ema.B[off : off + c].copy_(sd[k].reshape(-1))
                    off += c
            self.ema.updates = int(ckpt.get("updates") or 0)
        if isinstance(exact, dict) and exact.get("P") is not None and exact["P"].numel() == self.flat.P.numel():
            self.flat.P.copy_(exact["P"].to(dev))
            self.flat.B.copy_(exact["B"].to(dev))
            self.ema.P.copy_(exact["ema_P"].to(dev))
            self.ema.B.copy_(exact["ema_B"].to(dev))
            self.buf1.copy_(exact["buf1"].to(dev))
            if self.buf2 is not None and exact.get("buf2") is not None:
                self.buf2.copy_(exact["buf2"].to(dev))
            if self.amp_state is not None and exact.get("amp_state") is not None:
                self.amp_state.copy_(exact["amp_state"].to(dev))
            if exact.get("G") is not None:
                self.flat.G.copy_(exact["G"].to(dev))
            self.opt_steps, self.last_opt_step = int(exact["opt_steps"]), int(exact["last_opt_step"])
            self._resumed_exact = True
        assert start_epoch > 0, (f"{self.args['model']} training to {self.epochs} epochs is finished, nothing to resume.\n"
                                 f"Start a new training without resuming, i.e. YOLO('{self.args['model']}').train()")
        LOGGER.info(f"Resuming training {self.args['model']} from epoch {start_epoch + 1} to {self.epochs} total epochs")
        if self.epochs < start_epoch:
            LOGGER.info(f"the model has been trained for {ckpt['epoch']} epochs. Fine-tuning for {self.epochs} more epochs.")
            self.epochs += int(ckpt["epoch"])  # finetune additional epochs
            self.sched.epochs = self.epochs
        self.best_fitness = best_fitness
        self.start_epoch = start_epoch

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """``torch.optim``'s state-dict layout (what ``optimizer_state_dict`` writes, fp16 in the file: torch_utils.py:619-632) back into the
        flat moment buffers; entries follow the reference's parameter enumeration (biases, decay weights, norm weights)."""
        g0, g1, g2 = param_group_names(self.model, include_frozen=True)
        order = list(g2) + list(g0) + list(g1)
        state = sd.get("state", {})
        ran = None
        for i, k in enumerate(order):
            st = state.get(i, state.get(str(i)))
            if st is None or k not in self.flat.offsets:
                continue
            off, c = self.flat.offsets[k]
            if self.opt_name == "SGD":
                if st.get("momentum_buffer") is not None:
                    self.buf1[off : off + c].copy_(st["momentum_buffer"].reshape(-1).float())
            else:
                self.buf1[off : off + c].copy_(st["exp_avg"].reshape(-1).float())
                self.buf2[off : off + c].copy_(st["exp_avg_sq"].reshape(-1).float())
                ran = int(float(st["step"])) if "step" in st else ran
        if state:
            # SGD's first step seeds the momentum buffer with the gradient (first-step flag = opt_steps == 1): any restored state is past it
            self.opt_steps = ran if ran is not None else max(self.opt_steps, 1)

    # ---- setup -----------------------------------------------------------------------------------------------------------
    def _device_list(self) -> List[int]:
        dev = self.args.get("device", "")
        if dev in ("", None):
            return [self.local_rank]
        if isinstance(dev, (list, tuple)):
            return [int(x) for x in dev]
        return [int(x) for x in str(dev).replace("cuda:", "").split(",") if x.strip() != ""]

    def _setup_model_state(self) -> None:
        """Model onto this rank's GPU, flat buffers, optimizer state, EMA — the device part of _setup_train (trainer.py:231-317)."""
        a = self.args
        devs = self._device_list()
        index = self.local_rank if (self.world > 1 and "LOCAL_RANK" in os.environ) else devs[0]
        if os.environ.get("DYOLO_FORCE_DEVICE"):  # N-rank rehearsal on one GPU (gloo)
            index = int(os.environ["DYOLO_FORCE_DEVICE"])
        self.device = torch.device("cuda", index)
        torch.manual_seed(int(a.get("seed", 0)))  # trainer.py:121 init_seeds, BEFORE the model is built: two runs of one process start from the same weights
        if self.model is None:
            from ..nn.tasks import DetectionModel

            src = str(a.get("model") or "yolov8s-p2-repvgg.yaml")
            if src.endswith(".pt"):
                from ..nn.checkpoint import load_reference_checkpoint

                self.model, _ = load_reference_checkpoint(src)
            else:
                self.model = DetectionModel(src, nc=a.get("nc") or None, verbose=False)
        self.model = self.model.to(self.device).train()
        for k, v in self.model.named_parameters():  # _setup_train's freeze block (trainer.py:238-254): '.dfl' stays frozen, the rest trains
            v.requires_grad_(".dfl" not in k and v.dtype.is_floating_point)  # (a predictor may have frozen the graph before)
        self.model.train_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.get("dtype", "bf16")] if a.get("amp", True) \
            else torch.float32
        self.model.args = type("Args", (), dict(box=a["box"], cls=a["cls"], dfl=a["dfl"]))()
        self.batch_size = int(a["batch"])  # GLOBAL batch, as in the reference; each rank sees batch // world
        self.weight_decay = a["weight_decay"] * self.batch_size * max(round(a["nbs"] / max(self.batch_size, 1)), 1) / a["nbs"]  # trainer.py:254-256
        self.flat = FlatState(self.model, self.device)
        if P.exchange_on():
            # DDP broadcasts rank 0's parameters and buffers at construction; so do we (replicas must not rely on equal seeds)
            for t in (self.flat.P, self.flat.B):
                if torch.distributed.get_backend() == "nccl":
                    torch.distributed.broadcast(t, 0)
                else:  # gloo rehearsal on fewer GPUs than ranks: through host memory
                    h = t.cpu()
                    torch.distributed.broadcast(h, 0)
                    t.copy_(h)
        nc = self.model.yaml["nc"]
        name, self.lr0, self.momentum, self.warmup_bias_lr = resolve_optimizer(a, nc, self.iterations_hint)
        if name not in ("SGD", "AdamW"):
            raise NotImplementedError(f"optimizer {name}: SGD and AdamW are built")
        self.opt_name = name
        self.sched = OptimSchedule(a, name, self.lr0, self.momentum, self.warmup_bias_lr, self.batch_size, self.epochs)
        self.lf = self.sched.lf
        n = self.flat.P.numel()
        self.buf1 = torch.zeros(n, dtype=torch.float32, device=self.device)  # momentum / first moment
        self.buf2 = torch.zeros(n, dtype=torch.float32, device=self.device) if name == "AdamW" else None
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        # GradScaler(enabled=amp) of the reference (trainer.py:271) for fp16 storage: {scale, growth tracker, found_inf, skipped} on the
        # device, so a step never waits for the host.  bf16 / fp32 storage have fp32's exponent range: no scaler (scale stays 1).
        self.amp_state = torch.tensor([65536.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=self.device) if self.model.train_dtype == torch.float16 else None
        self.ema = ModelEMA(self.flat)
        self.opt_steps = 0
        self.iters = 0
        self.last_opt_step = -1
        # (a one-rank group under DYOLO_DDP_SINGLE_RANK=1 exchanges too: every RCCL call of the N-rank job on one GPU)
        self.buckets = P.GradBuckets(self.flat, n_buckets=int(os.environ.get("DYOLO_GRAD_BUCKETS", 4))) if P.exchange_on() else None
        if self.grad_sink:
            self.flat.enable_sink()
            if self.buckets is not None:
                self.buckets.use_sink()  # several ranks: the sink is flushed bucket by bucket, each flush followed by the bucket's all-reduce

    # ---- schedules (state lives in self.sched) ------------------------------------------------------------------------------
    cur_lrs = property(lambda self: self.sched.cur_lrs)
    cur_momentum = property(lambda self: self.sched.cur_momentum)
    accumulate = property(lambda self: self.sched.accumulate)

    def warmup_iters(self, nb: int) -> int:
        return self.sched.warmup_iters(nb)

    def scheduler_step(self, epoch: int) -> None:
        self.sched.scheduler_step(epoch)

    def warmup(self, ni: int, epoch: int, nb: int) -> None:
        self.sched.warmup(ni, epoch, nb)

    def lr_momentum(self, ni: int, epoch: int, nb: int) -> Tuple[List[float], float]:
        """Per-group lr and momentum the optimizer uses at batch counter ``ni`` of ``epoch`` (stateless view of the above)."""
        self.scheduler_step(epoch)
        self.warmup(ni, epoch, nb)
        return list(self.cur_lrs), self.cur_momentum

    # ---- one iteration ---------------------------------------------------------------------------------------------------
    def step(self, batch: Dict[str, torch.Tensor], epoch: int = 0, nb: int = 1000):
        """One batch outside the epoch loop (tests, bench): warm-up by this trainer's own batch counter, optimizer step when
        ``accumulate`` batches have been seen.  batch: img (N_local, 3, H, W) uint8/float on the device, batch_idx / cls /
        bboxes as the reference's collate gives them.  Returns (loss, loss_items) of this rank."""
        ni = self.iters
        self.scheduler_step(epoch)
        return self.train_batch(batch, ni, epoch, nb)

    def train_batch(self, batch: Dict[str, torch.Tensor], ni: int, epoch: int, nb: int):
        """Warm-up, forward, loss * world (folded into the SUM all-reduce), backward, optimizer step — trainer.py:362-399."""
        self.warmup(ni, epoch, nb)
        will_step = ni - self.last_opt_step >= self.accumulate
        if self.buckets is not None:
            self.buckets.arm(will_step)  # the backward that precedes an optimizer step all-reduces its buckets as they fill
        loss, items = self._forward_backward(batch)
        self.iters += 1
        if will_step:
            self.optimizer_step()
            self.last_opt_step = ni
            self._tick("optimizer_step")
        return loss.detach() * self.world, items  # the reference reports loss * world_size (trainer.py:382-383)

    graph_steps = os.environ.get("DYOLO_TRAIN_GRAPH", "1") != "0"  # forward + loss + backward recorded once as a hipGraph and replayed (see _forward_backward)
    grad_sink = True  # parameter gradients through FlatState's sink (one flush per batch / per bucket) instead of AccumulateGrad

    def _forward_backward(self, batch: Dict[str, torch.Tensor]):
        """loss, items = model(batch); loss.backward() (trainer.py:379-389).

        A step is ~2,400 kernel launches from Python; the GPU needs ~40 ms for them, the host 40-130 ms depending on what
        else runs on the box, so the step is host bound.  The whole forward + loss + backward is therefore captured into a
        hipGraph (torch.cuda.CUDAGraph: activations live in the graph's private pool) once per (image shape, label capacity)
        and replayed: the images and the label table are copied into static buffers, the parameter gradients accumulate into
        the flat gradient buffer exactly as in the eager backward.  Nothing in it depends on a host value that changes between
        steps; the optimizer step (learning rate, momentum, EMA decay, first-step flag) stays outside.
        Several ranks (reference: DistributedDataParallel, trainer.py:274, whose bucket all-reduces overlap backward): the capture is cut
        behind every gradient bucket's sink flush (``_capture_cut``): K graphs, bucket k's all-reduce issued between the launches of graph k
        and graph k + 1 — the ring runs under the backward kernels that follow, and the host still enqueues ~10 calls per step."""
        use = (self.graph_steps and self.iters >= 2 and batch["img"].is_cuda and self.model.training
               and (self.buckets is None or self.grad_sink) and not self.args.get("multi_scale"))  # (multi_scale: ~40 input shapes, a graph pool each: eager)
        from ..nn.autograd_ops import lazy_head_seed, sink_armed

        bk = self.buckets

        def backward(loss):
            # scaler.scale(loss).backward() (trainer.py:389): the seed of the backward pass is the device-resident scale
            with sink_armed(bk.note if bk is not None else None), lazy_head_seed():
                loss.backward(gradient=self.amp_state[0] if self.amp_state is not None else self._unit_seed(loss))
            if bk is not None:
                bk.end_backward()  # buckets backward did not run down (parameters without a gradient): flushed / issued in order
            else:
                self.flat.flush_sink()

        from .. import hip_ops as H

        packs = self.__dict__.get("_pack_cache")
        if packs is None and batch["img"].is_cuda and os.environ.get("DYOLO_PACK_BATCH", "1") != "0":
            packs = self._pack_cache = H.PackCache(self.model.train_dtype, batch["img"].device, self.flat.P)  # every layer's weights packed by ONE launch per step
        if not use:
            with H.batched_weight_packing(packs):
                if packs is not None:
                    packs.pack_all()
                with sink_armed():
                    loss, items = self.model(batch)
                backward(loss)
            # detached: a caller that keeps the loss must not keep the autograd graph alive — its AccumulateGrad nodes would stay bound to
            # this stream, and the next capture (another stream) then breaks inside hipStreamEndCapture (seen as a segfault on ROCm 7.0)
            return loss.detach(), items.detach()
        model = self.model
        if getattr(model, "criterion", None) is None:
            model.criterion = model.init_criterion()
        img = batch["img"]
        b, _, h, w = img.shape
        self._tick("outside_step")
        gt = model.criterion.targets_to_gt(batch, b, (h, w))
        # label capacity of the static table: multiples of 64, never shrinking for a shape (Poisson(50) counts wander across 64 from
        # batch to batch; re-capturing on every crossing was a silent cliff), one graph per (shape, capacity) kept in a small dict
        caps = self.__dict__.setdefault("_graph_caps", {})
        shape_key = (tuple(img.shape), img.dtype, model.train_dtype)
        cap = max(caps.get(shape_key, 128), -(-gt.shape[1] // 64) * 64)
        caps[shape_key] = cap
        key = shape_key + (cap,)
        graphs = self.__dict__.setdefault("_graphs", {})
        gs = graphs.get(key)
        if gs is None:
            for k in [k for k in graphs if k[:3] == shape_key]:  # a smaller capacity of this shape is never used again
                del graphs[k]
            gs = dict(key=key, img=torch.empty_like(img), gt=torch.zeros((b, cap, 5), dtype=torch.float32, device=img.device))
            gs["img"].copy_(img)
            armed = bk.armed if bk is not None else False
            if bk is not None:
                bk.arm(armed, capturing=True)

            def body():
                if packs is not None:
                    packs.pack_all()
                with sink_armed():
                    loss, items = model.criterion.from_gt(model.forward_train(gs["img"]), gs["gt"])
                backward(loss)
                gs.update(loss=loss, items=items)

            torch.cuda.synchronize(img.device)
            cut = bk is not None and os.environ.get("DYOLO_DDP_GRAPH_CUT", "1") != "0"
            if not cut:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g), H.batched_weight_packing(packs):
                    body()
                gs.update(graphs=[g], ends=[len(bk.buckets) if bk is not None else 0])
            else:
                gs.update(graphs=[], ends=[])
                self._capture_cut(gs, bk, packs, body)
            if bk is not None:
                bk.arm(armed)
            graphs[key] = gs
        self._graph = gs
        tick = self._tick
        tick("targets_to_gt")
        if img.data_ptr() != gs["img"].data_ptr():  # a caller that fills `static_image()` itself skips the device copy
            gs["img"].copy_(img, non_blocking=True)
        tick("image_copy")
        # The label table goes through a ring of PINNED host tables (zero rows = padding) and ONE asynchronous copy.  Round 4 copied a
        # pageable tensor with non_blocking=True: a pageable hipMemcpyAsync is stream-ordered but returns only when it has run, so the host
        # sat in it until the previous step's whole graph and optimizer had drained, and the next graph launch (~2,400 nodes of host work)
        # then started on an idle GPU: the launch cost was serial to every step instead of hidden behind the one before.
        ring = gs.get("gt_ring")
        if ring is None:
            ring = gs["gt_ring"] = dict(tables=[torch.zeros((b, cap, 5), dtype=torch.float32).pin_memory() for _ in range(3)], events=[None] * 3, at=0)
        k = ring["at"]
        ring["at"] = (k + 1) % 3
        if ring["events"][k] is not None:
            ring["events"][k].synchronize()  # the copy that last read this table (three steps ago)
        host = ring["tables"][k]
        hv = host.numpy()  # (numpy on purpose: see v8DetectionLoss.preprocess)
        hv.fill(0.0)
        if gt.shape[1]:
            hv[:, : gt.shape[1]] = gt.numpy()
        if os.environ.get("DYOLO_PAGEABLE_GT") == "1":  # round 4's form, kept for the A/B of profiles/r05_train_host_ab.txt only
            gs["gt"].copy_(host.clone().to(img.device, non_blocking=True))
        else:
            gs["gt"].copy_(host, non_blocking=True)
        ev = ring["events"][k] = ring["events"][k] or torch.cuda.Event()
        ev.record()
        tick("label_copy")
        if bk is not None:
            bk.begin_replay()
        for g, end in zip(gs["graphs"], gs["ends"]):
            g.replay()
            if bk is not None:
                bk.exchange_upto(end)  # the all-reduces of the buckets this graph flushed: they run under the graphs launched next
        tick("replay")
        out = gs["loss"].clone(), gs["items"].clone()  # the static outputs are overwritten by the next replay (the epoch mean keeps them)
        tick("exchange_and_outputs")
        return out

    def _capture_cut(self, gs: dict, bk, packs, body) -> None:
        """Several ranks (reference: DistributedDataParallel overlaps its bucket all-reduces with backward, trainer.py:274): the step is
        captured as K hipGraphs CUT behind each gradient bucket's sink flush — ``GradBuckets.cut`` ends the running capture and begins the
        next graph in the same memory pool — so that a replayed step is: launch graph 0, issue bucket 0's all-reduce, launch graph 1, ...
        RCCL's stream waits for what the compute stream held when the all-reduce was issued (graph k), and the ring runs under the backward
        kernels of graphs k + 1 ..: the eager form's overlap at a handful of host calls per step.  (r03 / r04 wanted ONE graph with an
        external event per bucket; this ROCm refuses external events.)  The cut happens inside ``loss.backward()``, i.e. on autograd's
        device thread, hence relaxed capture mode: begin and end of one capture may then sit on different threads."""
        import gc

        from .. import hip_ops as H

        dev = gs["img"].device
        gc.collect()
        torch.cuda.empty_cache()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        state = {"pool": torch.cuda.graph_pool_handle(), "open": None}  # ONE memory pool: what graph k allocates, graph k + 1 reads

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=state["pool"], capture_error_mode="relaxed")
            state["open"] = g

        def end(upto: int):
            g, state["open"] = state["open"], None
            g.capture_end()
            gs["graphs"].append(g)
            gs["ends"].append(upto)

        def cut(bi: int):  # called by GradBuckets right behind bucket bi's flush (autograd's thread)
            end(bi + 1)
            begin()

        with torch.cuda.stream(side), H.batched_weight_packing(packs):
            begin()
            bk.cut = cut
            try:
                body()
            finally:
                bk.cut = None
                if state["open"] is not None:
                    end(len(bk.buckets))
        torch.cuda.current_stream(dev).wait_stream(side)

    host_phases: Optional[Dict[str, float]] = None  # set to {} to collect the host's seconds per phase of the graphed step (bench.py)

    def _tick(self, name: str) -> None:
        """Host time since the previous tick goes to phase ``name`` (no device call; off unless ``host_phases`` is a dict)."""
        hp = self.host_phases
        if hp is None:
            return
        now = time.perf_counter()
        last = self.__dict__.get("_tick_last")
        if last is not None:
            hp[name] = hp.get(name, 0.0) + (now - last)
            worst = hp.setdefault("__max__", {})
            if now - last > worst.get(name, (0.0, 0))[0]:
                worst[name] = (now - last, self.iters)  # the longest single visit of the phase and the iteration it happened in
        self._tick_last = now

    def static_image(self) -> Optional[torch.Tensor]:
        """The graphed step's resident image batch (None before the first capture): a loader that writes the next batch here saves the device copy."""
        g = getattr(self, "_graph", None)
        return None if g is None else g["img"]

    def _unit_seed(self, loss: torch.Tensor) -> torch.Tensor:
        """ones_like(loss), allocated once (the seed autograd would create per call; a device fp32 scalar the head-gradient kernel reads)."""
        u = self.__dict__.get("_unit")
        if u is None or u.device != loss.device or u.dtype != loss.dtype or u.shape != loss.shape:
            u = self._unit = torch.ones_like(loss)
        return u

    def step_form(self) -> str:
        """How a step is issued (for the bench line): eager launches, one hipGraph, or hipGraphs cut at the gradient-bucket boundaries."""
        gs = getattr(self, "_graph", None)
        if self.buckets is None:
            return "forward + loss + backward replayed as ONE hipGraph, gradients through the sink (one flush)" if gs is not None else "eager launches"
        if gs is not None:
            k = len(gs["graphs"])
            if k > 1:
                return (f"forward + loss + backward replayed as {k} hipGraphs cut behind each gradient bucket's sink flush; bucket k's all-reduce is issued "
                        f"between the launches of graph k and graph k + 1 (overlaps the remaining backward)")
            return ("forward + loss + backward replayed as ONE hipGraph with a per-bucket sink flush; the bucket all-reduces are issued AFTER the graph "
                    "(no overlap with backward)")
        return "eager launches, bucket all-reduces issued from backward as each bucket's last gradient lands (overlap with the remaining backward)"

    def optimizer_step(self) -> None:
        """scaler.unscale_, clip 10, scaler.step, scaler.update, zero_grad, EMA — trainer.py:591-599.  The scaler exists for fp16
        storage only; unscale and the skip-on-overflow live inside the step kernels (``amp_state``), so nothing here reads the device."""
        from .. import hip_ops as H

        G, Pm = self.flat.G, self.flat.P
        if self.buckets is not None:
            self.buckets.finish()  # wait for the in-flight bucket all-reduces, reduce whatever backward did not reach
        self.sumsq.zero_()
        H.sumsq_into(self.sumsq, G)
        self.opt_steps += 1
        for gi, sl in enumerate(self.flat.group_slices()):
            if sl.stop == sl.start:
                continue
            wd = self.weight_decay if gi == 0 else 0.0
            if self.opt_name == "SGD":
                H.sgd_step_(Pm[sl], G[sl], self.buf1[sl], self.cur_lrs[gi], self.cur_momentum, wd, True, self.opt_steps == 1, self.sumsq, 10.0, self.amp_state)
            else:
                H.adamw_step_(Pm[sl], G[sl], self.buf1[sl], self.buf2[sl], self.cur_lrs[gi], (self.momentum, 0.999), 1e-8, wd, self.opt_steps, self.sumsq, 10.0,
                              self.amp_state)
        if self.amp_state is not None:
            H.amp_update_(self.amp_state, self.sumsq)
        G.zero_()
        self.ema.update()

    # ---- the loop --------------------------------------------------------------------------------------------------------
    def train(self):
        """trainer.py:171-207: launch one rank per GPU when several devices are asked for and we are not a rank yet."""
        devs = self._device_list()
        world = len(devs) if len(devs) > 1 else 1
        if world > 1 and "LOCAL_RANK" not in os.environ:
            from ..utils.dist import ddp_cleanup, generate_ddp_command, rank_env, visible_device_env, visible_gpu_count

            rehearsal = bool(os.environ.get("DYOLO_FORCE_DEVICE"))
            if not rehearsal and visible_gpu_count() < world:
                raise RuntimeError(f"device={self.args['device']!r} asks for {world} GPUs, this node has {visible_gpu_count()}")
            # local rank i must run on the i-th REQUESTED GPU (select_device exports CUDA_VISIBLE_DEVICES=device, torch_utils.py:183)
            env = rank_env(None if rehearsal else visible_device_env(devs))
            overrides = {k: v for k, v in self.args.items()}
            tmp_data = None
            if isinstance(overrides.get("data"), dict):  # tensors do not survive repr() into the rank script: hand them over as a file
                import tempfile

                tmp_data = os.path.join(tempfile.mkdtemp(prefix="dyolo_data_"), "data.pt")
                torch.save({k: v for k, v in overrides["data"].items()}, tmp_data)
                overrides["data"] = tmp_data
            cmd, file = generate_ddp_command(world, overrides)
            try:
                LOGGER.info(f"DDP: debug command {' '.join(cmd)}")
                subprocess.run(cmd, check=True, env=env)
            finally:
                ddp_cleanup(file)
                if tmp_data and os.path.exists(tmp_data):
                    os.remove(tmp_data)
                    os.rmdir(os.path.dirname(tmp_data))
            return None
        return self._do_train(world)

    def _setup_train(self, world: int) -> None:
        a = self.args
        nc_hint = int(a.get("nc") or (self.model.yaml["nc"] if self.model is not None else 10))
        data = load_dataset(a.get("data") or "synthetic", int(a["imgsz"]), nc_hint, int(a.get("seed", 0)))
        if self.flat is None:
            # build_optimizer's 'auto' rule looks at the planned number of iterations (trainer.py:309-310)
            self.iterations_hint = math.ceil(data["img"].shape[0] / max(int(a["batch"]), a["nbs"])) * self.epochs
            self._setup_model_state()
        per_rank = max(self.batch_size // max(world, 1), 1)  # trainer.py:286
        self.train_loader = TensorLoader(data, per_rank, self.rank if world > 1 else 0, max(world, 1), seed=int(a.get("seed", 0)))
        if self.rank == 0:
            self.wdir.mkdir(parents=True, exist_ok=True)
            if a.get("val", True):
                # _setup_train (trainer.py:288-296): the validation loader lives on rank 0 at twice the per-rank batch.  A tensor dataset
                # carries its split as data["val"] (same layout); without one the training tensors are evaluated
                from .validator import DetectionValidator

                # the reference needs a real val split (check_det_dataset).  Without one there is nothing to validate ON: the
                # metrics/* and val/* columns and best.pt stay out rather than being computed on the training tensors under a validation
                # label; `val: "train"` asks for exactly that evaluation by name (smoke runs, the self-consistency test of the validator)
                vd = data.get("val") if isinstance(data.get("val"), dict) else (data if a.get("val") == "train" else None)
                if vd is None:
                    LOGGER.info("val: the dataset has no 'val' split -- validation skipped (fitness = -loss decides best.pt)")
                else:
                    self.val_loader = TensorLoader({k: vd[k] for k in ("img", "batch_idx", "cls", "bboxes")}, per_rank * 2, 0, 1, shuffle=False)
                    self.validator = DetectionValidator(a)
        self.resume_training(self._resume_ckpt)  # trainer.py:315 (after the optimizer exists)
        self._resume_ckpt = None

    def _do_train(self, world: int = 1):
        if world > 1:
            P.init_distributed()  # _setup_ddp (trainer.py:218-229): RCCL, one rank per GPU
            self.rank, self.local_rank, self.world = P.dist_env()
        self._setup_train(world)
        nb = len(self.train_loader)
        nw = self.warmup_iters(nb)
        if not getattr(self, "_resumed_exact", False):
            self.last_opt_step = -1
            self.flat.G.zero_()
        t_start = self.train_time_start = time.time()
        self.stop = False
        LOGGER.info(f"Image sizes {self.args['imgsz']} train\\nLogging results to {self.save_dir}\\nStarting training for {self.epochs} epochs...") if self.rank == 0 else None
        epoch = self.start_epoch
        while True:
            self.epoch = epoch
            self.scheduler_step(epoch)
            self.model.train()
            self.train_loader.set_epoch(epoch)
            self.tloss = None
            for i, batch in enumerate(self.train_loader):
                ni = i + nb * epoch
                batch = self.preprocess_batch(batch)
                self.loss, self.loss_items = self.train_batch(batch, ni, epoch, nb)
                self.tloss = (self.tloss * i + self.loss_items) / (i + 1) if self.tloss is not None else self.loss_items
                if self.args.get("time"):  # timed stopping (trainer.py:396-404): rank 0's clock decides for every rank
                    self.stop = P.broadcast_flag((time.time() - t_start) > float(self.args["time"]) * 3600, self.device)
                    if self.stop:
                        break
            final_epoch = epoch + 1 >= self.epochs
            if self.rank == 0:
                self.lr = {f"lr/pg{ir}": x for ir, x in enumerate((self.cur_lrs[2], self.cur_lrs[0], self.cur_lrs[1]))}  # reference group order
                if self.validator is not None and (self.args.get("val", True) or final_epoch or self.stopper.possible_stop or self.stop):  # trainer.py:430-432
                    self.metrics, self.fitness = self.validate()
                elif self.validator is None:
                    self.fitness = -float(self.loss)  # (validate()'s own rule when there are no metrics, trainer.py:611-612)
                    if not self.best_fitness or self.best_fitness < self.fitness:
                        self.best_fitness = self.fitness
                self.save_metrics({"time": time.time() - t_start, **self.label_loss_items(self.tloss), **self.metrics, **self.lr})
                self.stop |= self.stopper(epoch + 1, self.fitness) or final_epoch  # trainer.py:435
                if self.args.get("time"):
                    self.stop |= (time.time() - t_start) > float(self.args["time"]) * 3600
                if self.args.get("save", True) or final_epoch:
                    self.save_model()
                LOGGER.info(f"{epoch + 1}/{self.epochs}  " + "  ".join(f"{k} {float(v):.4g}" for k, v in self.label_loss_items(self.tloss).items()))
            # trainer.py:457-463: rank 0's decision reaches every rank (all of them must leave the loop in the same epoch) — one int32 over RCCL
            self.stop = P.broadcast_flag(bool(self.stop) or final_epoch, self.device)
            if self.stop:
                break
            epoch += 1
        torch.cuda.synchronize(self.device)
        if self.rank == 0:
            LOGGER.info(f"{epoch - self.start_epoch + 1} epochs completed in {(time.time() - t_start) / 3600:.3f} hours.")
            self.final_eval()
        if world > 1 and torch.distributed.is_initialized():
            torch.distributed.barrier()
        return {**self.label_loss_items(self.tloss), "save_dir": str(self.save_dir)}

    def final_eval(self) -> None:
        """trainer.py:681-695: the optimizer (and this trainer's exact-resume state) stripped from last.pt and best.pt; best.pt takes
        last.pt's ``train_results``, is validated once more and its metrics become the run's."""
        from ..utils.torch_utils import strip_optimizer

        ckpt = {}
        for f in (self.last, self.best):
            if not f.exists():
                continue
            if f is self.last:
                ckpt = strip_optimizer(f)
            else:
                k = "train_results"
                best = strip_optimizer(f, updates={k: ckpt[k]} if k in ckpt else None)
                if self.validator is not None and best.get("model") is not None:
                    LOGGER.info(f"Validating {f}...")
                    sd = {k2: v.float() for k2, v in best["model"].state_dict().items() if v.is_floating_point()}
                    keep = (self.ema.P.clone(), self.ema.B.clone())
                    try:  # validate() evaluates the EMA buffers: lend them best.pt's weights for the pass
                        for k2, (off, c) in self.flat.offsets.items():
                            self.ema.P[off : off + c].copy_(sd[k2].reshape(-1))
                        off = 0
                        for k2, b in self.model.named_buffers():
                            if b.is_floating_point():
                                self.ema.B[off : off + b.numel()].copy_(sd[k2].reshape(-1))
                                off += b.numel()
                        bf = self.best_fitness
                        self.metrics, _ = self.validate()
                        self.best_fitness = bf
                    finally:
                        self.ema.P.copy_(keep[0])
                        self.ema.B.copy_(keep[1])

    def validate(self):
        """trainer.py:605-615 + validator.py:109-221 (training branch): the EMA weights evaluated on the validation tensors; fitness =
        0.1 mAP50 + 0.9 mAP50-95 (metrics.py ``Metric.fitness``), the best one remembered.  The EMA copy is swapped into the flat
        parameter / BatchNorm buffers for the pass (the module parameters are views of them) and the live weights put back after."""
        live_p, live_b = self.flat.P.clone(), self.flat.B.clone()
        self.flat.P.copy_(self.ema.P)
        self.flat.B.copy_(self.ema.B)
        try:
            metrics = self.validator(self.model, self.val_loader, self.device, self.model.train_dtype)
        finally:
            self.flat.P.copy_(live_p)
            self.flat.B.copy_(live_b)
            self.model.train()  # (drops the packs folded from the EMA weights)
        fitness = metrics.pop("fitness", None)
        if fitness is None:
            fitness = -float(self.loss)
        if not self.best_fitness or self.best_fitness < fitness:
            self.best_fitness = fitness
        return metrics, fitness

    def preprocess_batch(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """detect/train.py:57-74: images to the device (the /255 and the layout change ride in the first kernel); ``multi_scale``: the
        batch resized to a random multiple of the stride in [0.5, 1.5] x imgsz — same draw (``random.randrange``), same size rule, the
        resize itself = interpolate(img / 255, bilinear, align_corners=False) as one kernel on the uint8 batch."""
        batch["img"] = batch["img"].to(self.device, non_blocking=True)
        if self.args.get("multi_scale"):
            import random

            from .. import hip_ops as H

            imgs = batch["img"]
            stride = max(int(self.model.stride.max()), 32)
            imgsz = int(self.args["imgsz"])
            sz = random.randrange(int(imgsz * 0.5), int(imgsz * 1.5 + stride)) // stride * stride
            sf = sz / max(imgs.shape[2:])
            if sf != 1:
                ns = [math.ceil(x * sf / stride) * stride for x in imgs.shape[2:]]
                if imgs.dtype != torch.uint8:
                    raise NotImplementedError("multi_scale resizes the loader's uint8 batches (dy_resize_bilinear_u8_nchw_f32); float batches are not built")
                imgs = H.resize_bilinear_u8(imgs.contiguous(), ns)
            batch["img"] = imgs
        return batch

    def label_loss_items(self, loss_items=None, prefix: str = "train") -> Dict[str, float]:
        """detect/train.py:107-118."""
        keys = [f"{prefix}/{x}" for x in self.loss_names]
        if loss_items is None:
            return keys
        return dict(zip(keys, [round(float(x), 5) for x in loss_items]))

    def save_metrics(self, metrics: Dict[str, float]) -> None:
        """results.csv, one row per epoch — trainer.py:700-708."""
        keys, vals = list(metrics.keys()), list(metrics.values())
        new = not self.csv.exists()
        with open(self.csv, "a", newline="") as f:
            w = csv.writer(f)
            if new:
                w.writerow(["epoch"] + keys)
            w.writerow([self.epoch + 1] + [f"{v:.6g}" for v in vals])

    def read_results_csv(self) -> Dict[str, list]:
        if not self.csv.exists():
            return {}
        with open(self.csv) as f:
            rows = list(csv.DictReader(f))
        return {k: [float(r[k]) for r in rows] for k in (rows[0] if rows else {})}

    def optimizer_state_dict(self) -> dict:
        """``torch.optim`` state-dict layout of the reference's optimizer (param_groups: biases, decay weights, norm weights —
        trainer.py:810-819), built from the flat moment buffers."""
        g0, g1, g2 = param_group_names(self.model, include_frozen=True)  # the reference's enumeration: frozen parameters keep their index
        order = list(g2) + list(g0) + list(g1)
        state = {}
        # torch.optim.AdamW counts the steps that RAN: a GradScaler skips the optimizer on overflow (the reference saves no scaler state,
        # trainer.py:514-545), and the step kernels' bias correction uses opt_steps - skipped likewise (amp_state[3]; checkpoint-time host read)
        ran = self.opt_steps - (int(self.amp_state[3].item()) if self.amp_state is not None else 0)
        for i, k in enumerate(order):
            if k not in self.flat.offsets:
                continue  # frozen (dfl.conv.weight): never received a gradient, so torch.optim holds no state entry for it
            off, c = self.flat.offsets[k]
            shape = self.flat.params[k].shape
            if self.opt_name == "SGD":
                state[i] = {"momentum_buffer": self.buf1[off : off + c].view(shape).clone()}
            else:
                state[i] = {"step": torch.tensor(float(ran)), "exp_avg": self.buf1[off : off + c].view(shape).clone(),
                            "exp_avg_sq": self.buf2[off : off + c].view(shape).clone()}
        groups, s = [], 0
        for names, lr, wd in ((g2, self.cur_lrs[2], 0.0), (g0, self.cur_lrs[0], self.weight_decay), (g1, self.cur_lrs[1], 0.0)):
            grp = {"lr": lr, "initial_lr": self.lr0, "weight_decay": wd, "params": list(range(s, s + len(names)))}
            grp.update({"momentum": self.cur_momentum, "nesterov": True, "dampening": 0} if self.opt_name == "SGD" else {"betas": (self.momentum, 0.999), "eps": 1e-8})
            groups.append(grp)
            s += len(names)
        return {"state": state, "param_groups": groups}

    def save_model(self) -> None:
        """last.pt with the reference's keys (trainer.py:514-545): epoch, best_fitness, model None, ema = the EMA weights as a
        pickled fp16 module graph under the reference's class paths, updates, optimizer (fp16 state), train_args, ..."""
        from datetime import datetime

        from ..nn.checkpoint import save_reference_checkpoint

        args = {k: v for k, v in self.args.items() if not isinstance(v, dict)}  # (a tensor dataset passed as a dict does not belong in the file)
        # what the reference's keys cannot carry (they hold the EMA graph and the optimizer moments in fp16, and no live weights at all:
        # its resume restarts from the EMA): the fp32 state of the run, so that `resume` continues where this epoch ended.  Plain tensors
        # under a key of their own — the reference's loader ignores it, strip_optimizer drops it
        exact = {"P": self.flat.P.cpu(), "B": self.flat.B.cpu(), "ema_P": self.ema.P.cpu(), "ema_B": self.ema.B.cpu(), "buf1": self.buf1.cpu(),
                 "buf2": self.buf2.cpu() if self.buf2 is not None else None, "amp_state": self.amp_state.cpu() if self.amp_state is not None else None,
                 "G": self.flat.G.cpu() if self.accumulate > 1 else None, "opt_steps": self.opt_steps, "last_opt_step": self.last_opt_step, "iters": self.iters}
        save_reference_checkpoint(self.last, self.model, self.ema.state_dict(self.model), extra={
            "epoch": self.epoch, "best_fitness": self.best_fitness, "updates": self.ema.updates, "optimizer": self.optimizer_state_dict(),
            "train_args": args, "train_metrics": {**self.metrics, "fitness": self.fitness},
            "train_results": self.read_results_csv(), "date": datetime.now().isoformat(), "dyolo_state": exact})
        if self.best_fitness is not None and self.best_fitness == self.fitness:  # trainer.py:541-542
            import shutil

            shutil.copyfile(self.last, self.best)
