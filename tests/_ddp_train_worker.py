"""Rank program of tests/test_train_gpu.py::test_two_rank_graphed_steps_equal_eager_steps (not collected).

Started by drone_yolo_amd.utils.dist.launch_ranks under torch.distributed.run: two ranks that share ONE GPU (DYOLO_FORCE_DEVICE=0,
gloo for the exchange: RCCL refuses two ranks on a device) run K steps of the REAL DetectionTrainer on Drone-YOLO-n 64x64 —
gradient sink flushed per bucket, bucket all-reduces, and (DYOLO_TRAIN_GRAPH=1) forward + loss + backward replayed from hipGraphs cut at the
bucket boundaries (DYOLO_DDP_GRAPH_CUT=0: one graph, exchange behind it).
Also the rank program of test_one_rank_rccl_group_runs_the_exchange_path: ONE rank, backend nccl (= RCCL), DYOLO_DDP_SINGLE_RANK=1.  Rank 0 writes the parameters and what the trainer reports about its step form to argv[1]."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import drone_yolo_amd as D  # noqa: E402
from drone_yolo_amd import parallel as P  # noqa: E402
from drone_yolo_amd.engine.trainer import DetectionTrainer, synthetic_dataset  # noqa: E402
from drone_yolo_amd.utils.parity import seeded_state_dict  # noqa: E402


def main(out, steps=6, per_rank=4):
    rank, local_rank, world = P.init_distributed()
    dev = torch.device("cuda", int(os.environ.get("DYOLO_FORCE_DEVICE", local_rank)))
    torch.cuda.set_device(dev)
    model = D.DetectionModel("yolov8n-p2-repvgg.yaml", nc=10, verbose=False)
    model.load_state_dict(seeded_state_dict(model.state_dict(), 5, cls_bias=-1.6))
    tr = DetectionTrainer(model, dict(optimizer="SGD", lr0=0.01, momentum=0.9, batch=per_rank * world, nbs=per_rank * world, dtype="fp32", warmup_epochs=0.0))
    data = synthetic_dataset(per_rank * steps, 64, seed=100 + rank)
    t_host = 0.0
    losses = []
    for it in range(steps):
        sel = torch.arange(it * per_rank, (it + 1) * per_rank)
        rows = torch.isin(data["batch_idx"].long(), sel)
        batch = dict(img=data["img"][sel].to(dev), batch_idx=data["batch_idx"][rows] - it * per_rank, cls=data["cls"][rows], bboxes=data["bboxes"][rows])
        t0 = time.perf_counter()
        loss, _ = tr.step(batch)
        t_host += time.perf_counter() - t0
        losses.append(float(loss))
    torch.cuda.synchronize()
    psum = float(tr.flat.P.double().sum())
    same = P.max_over_ranks(psum, dev) == -P.max_over_ranks(-psum, dev)  # replicas identical (device tensors under RCCL, host tensors under gloo)
    ranks_sum = P.sum_over_ranks(1.0, dev)
    if rank == 0:
        torch.save({"P": tr.flat.P.cpu(), "losses": losses, "step_form": tr.step_form(), "graphs": len(tr._graph["graphs"]) if getattr(tr, "_graph", None) else 0,
                    "graphed": getattr(tr, "_graph", None) is not None, "replicas_identical": bool(same), "host_ms_per_step": t_host / steps * 1e3,
                    "buckets": len(tr.buckets.buckets), "world": world, "backend": torch.distributed.get_backend(), "ranks_sum": ranks_sum,
                    "issued_during_backward": tr.buckets.issued_during_backward}, out)
    torch.distributed.barrier()


if __name__ == "__main__":
    main(sys.argv[1])
