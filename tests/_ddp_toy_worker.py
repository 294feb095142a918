"""Rank program of tests/test_dist_gloo.py::test_launcher_runs_bucketed_data_parallel_steps (not collected).

Started by drone_yolo_amd.utils.dist.launch_ranks under torch.distributed.run with CPU ranks over gloo: a toy conv net
trained for a few steps with the trainer's own machinery — FlatState (flat parameter / gradient buffers), GradBuckets
(bucketed all-reduce issued from backward hooks), OptimSchedule, TensorLoader (DistributedSampler sharding) — and plain
momentum SGD on the flat buffers standing in for the dy_sgd_step kernel.  Rank 0 writes what it saw to argv[1]."""
import json
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from drone_yolo_amd import parallel as P  # noqa: E402
from drone_yolo_amd.engine.trainer import FlatState, OptimSchedule, TensorLoader, get_cfg, synthetic_dataset  # noqa: E402


def make_model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.SiLU(), nn.Conv2d(8, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.SiLU(),
                         nn.Conv2d(8, 4, 1, bias=True))


def loss_of(model, batch):
    x = batch["img"].float() / 255
    return (model(x) ** 2).sum() / 64.0  # a SUM over the images of the local batch, like v8DetectionLoss's loss * batch


def run(world_batch, steps, rank, world, buckets_n):
    model = make_model().eval()  # BatchNorm on its running statistics: per-rank batch statistics are not what is under test here
    if world > 1 and rank != 0:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)  # replicas start DIFFERENT on purpose: the broadcast below must fix it
    flat = FlatState(model, "cpu")
    if world > 1:
        dist.broadcast(flat.P, 0)
        dist.broadcast(flat.B, 0)
    a = get_cfg(dict(optimizer="SGD", batch=world_batch, epochs=2, warmup_epochs=0.0))
    sched = OptimSchedule(a, "SGD", 0.01, 0.9, 0.1, world_batch, 2)
    buckets = P.GradBuckets(flat, n_buckets=buckets_n) if world > 1 else None
    data = synthetic_dataset(16, 8, seed=4)
    loader = TensorLoader(data, world_batch // world, rank, world, seed=1, shuffle=False)
    mom = torch.zeros_like(flat.P)
    issued, done = [], 0
    for epoch in range(2):
        sched.scheduler_step(epoch)
        loader.set_epoch(epoch)
        for batch in loader:
            if done == steps:
                break
            if buckets is not None:
                buckets.arm(True)
            loss_of(model, batch).backward()
            if buckets is not None:
                issued.append(buckets.issued_during_backward)
                buckets.finish()
            mom.mul_(sched.cur_momentum).add_(flat.G)
            flat.P.add_(mom, alpha=-sched.cur_lrs[0])
            flat.G.zero_()
            done += 1
    return flat, issued, buckets


def main():
    out_path = sys.argv[1]
    rank, local_rank, world = P.init_distributed(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    flat, issued, buckets = run(8, 3, rank, world, buckets_n=3)
    # every rank must hold the same parameters after the steps
    mine = flat.P.clone()
    other = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    same = all(torch.equal(o, other[0]) for o in other)
    if rank == 0:
        json.dump({"same_on_all_ranks": bool(same), "issued_during_backward": issued, "n_buckets": len(buckets.buckets),
                   "bucket_numel": [b["numel"] for b in buckets.buckets], "params": flat.P.tolist(), "ranks": world}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
