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


def test_launcher_runs_bucketed_data_parallel_steps(tmp_path):
    """End to end through the rank launcher (utils/dist.py::launch_ranks -> python -m torch.distributed.run, 2 CPU ranks over
    gloo): a toy conv net takes three optimizer steps on the trainer's FlatState with the gradient exchange done by
    GradBuckets (three buckets, issued from backward hooks, in order).  Expectations: replicas that started different are
    made equal by the initial broadcast and stay equal; at least one bucket is all-reduced before backward has finished
    (the overlap); the result equals ONE process taking the same steps on the whole batches (sum of per-rank gradients)."""
    import json
    import sys

    from drone_yolo_amd.utils.dist import launch_ranks

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "ddp.json"
    rc = launch_ranks(2, os.path.join(root, "tests", "_ddp_toy_worker.py"), [str(out)], env={"DYOLO_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"},
                      allow_cpu_ranks=True)
    assert rc == 0 and out.exists()
    r = json.load(open(out))
    assert r["ranks"] == 2 and r["same_on_all_ranks"]
    assert r["n_buckets"] == 3 and sum(r["bucket_numel"]) == 3 * 8 * 9 + 8 * 8 * 9 + 8 * 4 + 16 + 16 + 4
    assert all(1 <= n <= 3 for n in r["issued_during_backward"]), r["issued_during_backward"]  # buckets left during backward, not after it
    sys.path.insert(0, os.path.join(root, "tests"))
    import _ddp_toy_worker as W

    model_steps = 3
    # single process: batches of 8 = the union of the two ranks' batches of 4 (rank r takes indices r::2, no shuffle)
    flat1, _, _ = W.run(8, model_steps, 0, 1, buckets_n=1)
    got = torch.tensor(r["params"])
    assert got.shape == flat1.P.shape and torch.isfinite(got).all()
    assert torch.allclose(got, flat1.P, rtol=1e-5, atol=1e-6), float((got - flat1.P).abs().max())  # SUM of the ranks' gradients == whole-batch gradient


def test_launcher_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` / `YOLO.train(device="0,1,..")` on a node with fewer GPUs must fail loudly, before starting anything."""
    import pytest

    from drone_yolo_amd.utils.dist import launch_ranks

    if torch.cuda.device_count() >= 64:
        pytest.skip("this node really has 64 GPUs")
    with pytest.raises(RuntimeError, match="GPU"):
        launch_ranks(64, "nonexistent.py")
