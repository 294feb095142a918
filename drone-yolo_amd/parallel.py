"""Multi-GPU layer: one process per GPU, `torch.distributed` over RCCL (backend "nccl") on xGMI.

Inference (the metric path) shards by image: rank r takes a contiguous slice of the batch, runs the
whole pass on its own GPU and exchanges nothing — the reference itself is single-device for predict
(`select_device('0,1')` still yields cuda:0, utils/torch_utils.py:202-219).  The only collectives
are control-plane: a barrier and a MAX-reduce of the elapsed time for measurement.
Training is data parallel as in the reference (engine/trainer.py:274, 286, 382-389): per-rank batch, loss * world_size
followed by DDP's mean, i.e. a SUM all-reduce of the gradients — ``allreduce_gradients`` on the trainer's flat buffer.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def dist_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group when launched by torch.distributed.run; "nccl" is RCCL."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # DYOLO_DIST_BACKEND=gloo: rehearse an N-rank run on fewer GPUs than ranks (RCCL refuses two ranks on one device)
            backend = os.environ.get("DYOLO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `n` items owned by `rank`; earlier ranks take the remainder."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_batch(batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    s, e = shard_range(batch.shape[0], rank, world)
    return batch[s:e]


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (RCCL needs a device tensor, gloo a CPU one)."""
    if not dist.is_initialized():
        return value
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_gradients(flat_grad: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce of the flat gradient buffer over all ranks (reference: loss *= world_size, trainer.py:382-383,
    then DistributedDataParallel's gradient mean — the product is the plain sum).  One bucket: for Drone-YOLO-s 43 MB fp32,
    a single ring pass over the xGMI links; BatchNorm statistics stay per rank (no SyncBN in the reference)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if flat_grad.is_cuda and dist.get_backend() != "nccl":  # gloo rehearsal: through host memory
            h = flat_grad.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat_grad.copy_(h)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


class GradBuckets:
    """Bucketed, overlapped gradient exchange on the trainer's flat fp32 gradient buffer.

    The parameters are cut, in REVERSE registration order (the order backward produces their gradients: Detect head first,
    stem last), into ``n_buckets`` runs of about equal size.  The flat buffer is group-major (decay weights | norm weights |
    biases), each group in registration order, so a run of consecutive parameters is one contiguous slice in each group: a
    bucket = up to three slices.  A post-accumulate hook on every parameter counts its bucket down; when the last gradient of
    a bucket has landed, the bucket's slices are SUM-all-reduced asynchronously (RCCL runs on its own stream behind an event
    on the compute stream, so it waits for the kernels that wrote those gradients and overlaps the rest of backward).
    Buckets are issued strictly in order on every rank (collectives must match across ranks); ``finish()`` issues whatever
    backward did not complete and waits for all of them.  xGMI is point-to-point: a ring all-reduce moves 2(N-1)/N of the bytes
    over each link, so a few ~10 MB buckets keep the links busy without paying the per-collective latency 238 times.
    Reference: DistributedDataParallel's bucketed all-reduce behind ``trainer.py:274``; loss * world_size (trainer.py:382-383)
    followed by DDP's mean is the plain SUM used here."""

    def __init__(self, flat, n_buckets: int = 4):
        self.flat = flat
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        bounds, off = [], 0
        for s in flat.sizes:
            bounds.append((off, off + s))
            off += s
        group_of = {k: gi for gi, g in enumerate(flat.groups) for k in g}
        order = [k for k in flat.reg_order][::-1]
        total = sum(flat.offsets[k][1] for k in order)
        n_buckets = max(1, min(n_buckets, len(order)))
        target = total / n_buckets
        self.buckets: List[dict] = []
        cur, acc = [], 0
        for k in order:
            cur.append(k)
            acc += flat.offsets[k][1]
            if acc >= target * (len(self.buckets) + 1) and len(self.buckets) < n_buckets - 1:
                self.buckets.append({"names": cur})
                cur = []
        if cur:
            self.buckets.append({"names": cur})
        for bi, b in enumerate(self.buckets):
            rng = {}
            for k in b["names"]:
                o, c = flat.offsets[k]
                lo, hi = rng.get(group_of[k], (o, o + c))
                rng[group_of[k]] = (min(lo, o), max(hi, o + c))
            b["ranges"] = [rng[g] for g in sorted(rng)]
            b["numel"] = sum(hi - lo for lo, hi in b["ranges"])
            assert b["numel"] == sum(flat.offsets[k][1] for k in b["names"]), "a bucket must be contiguous within each group"
            for k in b["names"]:
                flat.params[k].register_post_accumulate_grad_hook(lambda p, bi=bi: self._ready(bi))
        self.armed = False
        self.pending: List[int] = []
        self.next = 0
        self.works: list = []
        self.issued_during_backward = 0

    def arm(self, on: bool) -> None:
        """Call before a backward: ``on`` when that backward is followed by the optimizer step (the last micro-batch of an
        accumulation window) — earlier micro-batches only accumulate locally."""
        self.armed = bool(on) and self.world > 1
        self.pending = [len(b["names"]) for b in self.buckets]
        self.next, self.works, self.issued_during_backward = 0, [], 0

    def _issue(self, bi: int) -> None:
        for lo, hi in self.buckets[bi]["ranges"]:
            g = self.flat.G[lo:hi]
            if g.is_cuda and dist.get_backend() != "nccl":
                # rehearsal of N ranks on fewer GPUs (DYOLO_DIST_BACKEND=gloo): the exchange goes through host memory
                h = g.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                g.copy_(h)
            else:
                self.works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))

    def _ready(self, bi: int) -> None:
        if not self.armed:
            return
        self.pending[bi] -= 1
        while self.next < len(self.buckets) and self.pending[self.next] <= 0:  # in order, identically on every rank
            self._issue(self.next)
            self.next += 1
            self.issued_during_backward += 1

    def finish(self) -> None:
        if self.world <= 1:
            return
        if not self.armed:  # a step without an armed backward (should not happen): one plain all-reduce
            dist.all_reduce(self.flat.G, op=dist.ReduceOp.SUM)
            return
        while self.next < len(self.buckets):  # parameters that received no gradient this step
            self._issue(self.next)
            self.next += 1
        for w in self.works:
            w.wait()
        self.works, self.armed = [], False


def gather_detections(rows: torch.Tensor, counts: torch.Tensor) -> Optional[List[Tuple[torch.Tensor, torch.Tensor]]]:
    """Optional host-side gather of the small (n, max_det, 6)/(n,) results to rank 0."""
    if not dist.is_initialized():
        return [(rows.cpu(), counts.cpu())]
    obj = (rows.cpu(), counts.cpu())
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out
