"""GPU box: two-rank training rehearsal on one GPU (gloo), printing per-batch losses of each rank."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(DYOLO_FORCE_DEVICE="0", DYOLO_DIST_BACKEND="gloo")

if "RANK" in os.environ:
    import torch

    from drone_yolo_amd.engine.trainer import DetectionTrainer

    tr = DetectionTrainer(overrides=dict(model="yolov8n-p2-repvgg.yaml", data="synthetic:16", epochs=1, imgsz=64, batch=8, nbs=8, device="0,1", dtype="fp32", optimizer="SGD",
                                         warmup_epochs=0.0, project=tempfile.mkdtemp(), name="dbg"))
    orig = tr.train_batch

    def tb(batch, ni, epoch, nb):
        loss, items = orig(batch, ni, epoch, nb)
        torch.cuda.synchronize()
        print(f"rank {tr.rank} ni {ni} loss {float(loss):.4f} items {[round(float(v), 4) for v in items]} labels {batch['cls'].shape[0]} img {tuple(batch['img'].shape)} "
              f"issued {tr.buckets.issued_during_backward if tr.buckets else None} gnorm {float(tr.sumsq.sqrt()):.4f}", flush=True)
        return loss, items

    tr.train_batch = tb
    print(tr.train(), flush=True)
else:
    from drone_yolo_amd.utils.dist import launch_ranks

    sys.exit(launch_ranks(2, os.path.abspath(__file__), allow_cpu_ranks=True))
