"""CPU oracle of ONE training step of the Drone-YOLO path — TEST INFRASTRUCTURE ONLY (tests/, oracle/make_golden.py).

Restates, with plain PyTorch CPU ops and autograd, what the reference trainer does per batch:
  DetectionTrainer.preprocess_batch  (models/yolo/detect/train.py:57-74): img.float() / 255
  BaseModel.forward(dict) -> loss    (nn/tasks.py:98-114, 280-292): criterion(model(img), batch)
  loss.backward(); clip_grad_norm_(10.0); optimizer.step(); ema.update()  (engine/trainer.py:381-389, 591-599)
  build_optimizer                     (engine/trainer.py:764-825): three parameter groups
  ModelEMA.update                     (utils/torch_utils.py:515-545)
The module forward is oracle/drone_yolo_oracle.forward(fused="train"); the loss is oracle/loss_oracle.v8_detection_loss.
Pinned against the real reference by oracle/make_golden.py::train_vectors (tests/golden/train.npz).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from . import drone_yolo_oracle as O
from . import loss_oracle as LO

Tensor = torch.Tensor


def is_param(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var") or key.endswith("num_batches_tracked") or "dfl.conv" in key)


def loss_and_grads(d: dict, sd: Dict[str, Tensor], img_u8: Tensor, labels: Dict[str, Tensor]):
    """One forward/backward. ``sd`` is cloned: returns (total, items[3], grads{name: tensor}, new_sd with updated BN buffers)."""
    sd = {k: v.clone().float() if v.is_floating_point() else v.clone() for k, v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if is_param(k) and v.is_floating_point()}
    img = img_u8.float() / 255  # train.py:59
    feats = O.forward(d, sd, img, fused="train")
    layers = O.resolve_layers(d, img.shape[1])
    strides = O.model_strides(layers)
    nc = layers[-1][3][0]
    total, items = LO.v8_detection_loss(feats, labels, strides, nc)
    total.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in params.items()}
    return total.detach(), items, grads, {k: v.detach() for k, v in sd.items()}


def param_groups(sd: Dict[str, Tensor]) -> Tuple[List[str], List[str], List[str]]:
    """build_optimizer's split (trainer.py:795-808): g0 = weights with decay, g1 = norm-layer weights (no decay),
    g2 = biases (no decay).  Names: '...bn.weight' / '...bn.bias' are BatchNorm; any other '.bias' is a conv bias."""
    g0, g1, g2 = [], [], []
    for k in sd:
        if not is_param(k):
            continue
        if k.endswith(".bias"):
            g2.append(k)
        elif ".bn." in k or "rbr_identity" in k:
            g1.append(k)
        else:
            g0.append(k)
    return g0, g1, g2


def clip_grad_norm_(grads: Dict[str, Tensor], max_norm: float = 10.0) -> float:
    """torch.nn.utils.clip_grad_norm_ (trainer.py:594): scale by max_norm / (total_norm + 1e-6), clamped to 1."""
    total = math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values()))
    coef = min(max_norm / (total + 1e-6), 1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def sgd_step(sd, grads, bufs, lr: float, momentum: float, weight_decay: float, nesterov: bool = True):
    """torch.optim.SGD(nesterov=True) over the three groups (trainer.py:812-820): decay on g0 only."""
    g0, g1, g2 = param_groups(sd)
    for names, wd in ((g0, weight_decay), (g1, 0.0), (g2, 0.0)):
        for k in names:
            g = grads[k] + wd * sd[k] if wd else grads[k].clone()
            if k not in bufs:
                bufs[k] = g.clone()
            else:
                bufs[k].mul_(momentum).add_(g)
            g = g + momentum * bufs[k] if nesterov else bufs[k]
            sd[k] = sd[k] - lr * g


def adamw_step(sd, grads, state, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
    """torch.optim.AdamW over the three groups (decay on g0 only), decoupled decay p *= 1 - lr*wd."""
    g0, g1, g2 = param_groups(sd)
    for names, wd in ((g0, weight_decay), (g1, 0.0), (g2, 0.0)):
        for k in names:
            st = state.setdefault(k, {"step": 0, "m": torch.zeros_like(sd[k]), "v": torch.zeros_like(sd[k])})
            st["step"] += 1
            p = sd[k] * (1 - lr * wd)
            st["m"] = st["m"] * betas[0] + (1 - betas[0]) * grads[k]
            st["v"] = st["v"] * betas[1] + (1 - betas[1]) * grads[k] * grads[k]
            bc1, bc2 = 1 - betas[0] ** st["step"], 1 - betas[1] ** st["step"]
            sd[k] = p - (lr / bc1) * st["m"] / ((st["v"].sqrt() / math.sqrt(bc2)) + eps)


def ema_update(ema_sd, sd, updates: int, decay: float = 0.9999, tau: float = 2000.0) -> int:
    """ModelEMA.update (torch_utils.py:531-545): d = decay * (1 - exp(-updates / tau)); every floating entry of the
    state dict (parameters AND BatchNorm buffers) moves: v = d*v + (1-d)*model_v."""
    updates += 1
    dcy = decay * (1 - math.exp(-updates / tau))
    for k, v in ema_sd.items():
        if v.is_floating_point():
            ema_sd[k] = v * dcy + (1 - dcy) * sd[k].detach()
    return updates


def reference_schedule(args: dict, nb: int, batch_size: int, epochs: int, iterations: float, nc: int = 10):
    """The reference's optimizer construction and per-batch schedule, restated around REAL ``torch.optim`` objects: build_optimizer
    (engine/trainer.py:764-825: 'auto' rule, group order biases / decay weights / norm weights), ``_setup_scheduler``
    (:209-216, LambdaLR with ``last_epoch = start_epoch - 1``, :317) and the loop of ``_do_train`` (:345-390: ``scheduler.step()``
    per epoch, warm-up of accumulate / lr / momentum by ``ni = i + nb * epoch``, optimizer step when ``ni - last_opt_step >=
    accumulate``).  Returns one row per batch: (ni, lr of [biases, decay weights, norm weights], momentum or None, beta1 or
    None, accumulate, stepped)."""
    import numpy as np
    from torch import optim

    a = dict(args)
    name, lr, momentum = a["optimizer"], a["lr0"], a["momentum"]
    if name == "auto":
        lr_fit = round(0.002 * 5 / (4 + nc), 6)
        name, lr, momentum = ("SGD", 0.01, 0.9) if iterations > 10000 else ("AdamW", lr_fit, 0.9)
        a["warmup_bias_lr"] = 0.0
    g = [[torch.nn.Parameter(torch.zeros(1))] for _ in range(3)]  # decay weights, norm weights, biases
    accumulate = max(round(a["nbs"] / batch_size), 1)
    decay = a["weight_decay"] * batch_size * accumulate / a["nbs"]
    if name == "AdamW":
        opt = optim.AdamW(g[2], lr=lr, betas=(momentum, 0.999), weight_decay=0.0)
    else:
        opt = optim.SGD(g[2], lr=lr, momentum=momentum, nesterov=True)
    opt.add_param_group({"params": g[0], "weight_decay": decay})
    opt.add_param_group({"params": g[1], "weight_decay": 0.0})
    lf = lambda x: max(1 - x / epochs, 0) * (1.0 - a["lrf"]) + a["lrf"]  # noqa: E731
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scheduler = optim.lr_scheduler.LambdaLR(opt, lr_lambda=lf)
        scheduler.last_epoch = -1
        nw = max(round(a["warmup_epochs"] * nb), 100) if a["warmup_epochs"] > 0 else -1
        last_opt_step, rows = -1, []
        for epoch in range(epochs):
            scheduler.step()
            for i in range(nb):
                ni = i + nb * epoch
                if ni <= nw:
                    xi = [0, nw]
                    accumulate = max(1, int(np.interp(ni, xi, [1, a["nbs"] / batch_size]).round()))
                    for j, x in enumerate(opt.param_groups):
                        x["lr"] = np.interp(ni, xi, [a["warmup_bias_lr"] if j == 0 else 0.0, x["initial_lr"] * lf(epoch)])
                        if "momentum" in x:
                            x["momentum"] = np.interp(ni, xi, [a["warmup_momentum"], a["momentum"]])
                stepped = ni - last_opt_step >= accumulate
                if stepped:
                    last_opt_step = ni
                pg = opt.param_groups
                rows.append((ni, [float(x["lr"]) for x in pg], float(pg[0]["momentum"]) if "momentum" in pg[0] else None,
                             float(pg[0]["betas"][0]) if "betas" in pg[0] else None, accumulate, stepped))
    return name, decay, rows
