"""CPU, world_size 2 over gloo: the multi-GPU layer (batch split, max-over-ranks timing, result gather)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from drone_yolo_amd import parallel as P

    r, lr, w = P.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    batch = torch.arange(7 * 3).view(7, 3)
    mine = P.shard_batch(batch, r, w)
    P.barrier()
    tmax = P.max_over_ranks(1.0 + rank)
    tsum = P.sum_over_ranks(float(mine.shape[0]))
    rows = torch.full((mine.shape[0], 4, 6), float(rank))
    counts = torch.full((mine.shape[0],), rank + 1, dtype=torch.int32)
    gathered = P.gather_detections(rows, counts)
    if rank == 0:
        q.put((tmax, tsum, [tuple(g[0].shape) for g in gathered], [g[1].tolist() for g in gathered], mine.tolist()))
    dist.destroy_process_group()


def test_two_rank_batch_split_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, tsum, shapes, counts, mine0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0 and tsum == 7.0  # max over ranks of the step time; shards cover the batch exactly once
    assert shapes == [(4, 4, 6), (3, 4, 6)] and counts == [[1] * 4, [2] * 3]
    assert mine0 == torch.arange(12).view(4, 3).tolist()


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.nn as nn

    from drone_yolo_amd import parallel as P
    from drone_yolo_amd.engine.trainer import FlatState, param_group_names

    P.init_distributed(backend="gloo")
    torch.manual_seed(0)  # identical replicas
    model = nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Conv2d(8, 4, 1, bias=True))
    g0, g1, g2 = param_group_names(model)
    flat = FlatState(model, "cpu")
    # per-rank gradients as a data-parallel step would leave them: rank r contributes (r + 1) * ones
    for p in model.parameters():
        p.grad.add_(float(rank + 1))
    P.allreduce_gradients(flat.G)
    ok_views = all(p.grad.data_ptr() >= flat.G.data_ptr() for p in model.parameters())
    if rank == 0:
        q.put((g0, g1, g2, flat.sizes, float(flat.G.min()), float(flat.G.max()), ok_views, float(model[0].weight.grad.mean())))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_gloo():
    """The training exchange step: flat gradient buffer, parameters as views, SUM all-reduce over 2 ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g0, g1, g2, sizes, gmin, gmax, ok_views, wmean = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert g0 == ["0.weight", "2.weight"] and g1 == ["1.weight"] and g2 == ["1.bias", "2.bias"]
    assert sizes == [3 * 8 * 9 + 8 * 4, 8, 8 + 4]
    assert gmin == 3.0 and gmax == 3.0 and ok_views and wmean == 3.0  # 1 + 2 summed on every element, seen through the views
