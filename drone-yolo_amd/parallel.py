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
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def gather_detections(rows: torch.Tensor, counts: torch.Tensor) -> Optional[List[Tuple[torch.Tensor, torch.Tensor]]]:
    """Optional host-side gather of the small (n, max_det, 6)/(n,) results to rank 0."""
    if not dist.is_initialized():
        return [(rows.cpu(), counts.cpu())]
    obj = (rows.cpu(), counts.cpu())
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out
