"""A/B of the training step's label path (r05): the numpy label table against round 4's torch-CPU one, same process shape
as `bench.py --mode train`.  Prints the train record of each with the host phase maxima and the CFS throttle counters of the timed steps.

    python tools/train_label_ab.py [steps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from drone_yolo_amd.utils import loss as L  # noqa: E402
from drone_yolo_amd.utils.ops import xywh2xyxy  # noqa: E402


def torch_label_table(targets, batch_size, scale_tensor):
    """v8DetectionLoss.preprocess as round 4 had it: torch CPU ops (bincount / argsort / cumsum / index_put / xywh2xyxy)."""
    targets = targets.detach().cpu().float()
    nl, ne = targets.shape
    if nl == 0:
        return torch.zeros(batch_size, 0, ne - 1)
    img = targets[:, 0].long()
    counts = torch.bincount(img, minlength=batch_size)
    order = torch.argsort(img, stable=True)
    start = torch.cumsum(counts, 0) - counts
    pos = torch.arange(nl) - start[img[order]]
    out = torch.zeros(batch_size, int(counts.max()), ne - 1)
    out[img[order], pos] = targets[order, 1:]
    out[..., 1:5] = xywh2xyxy(out[..., 1:5] * scale_tensor)
    return out


def run(tag, steps):
    dev = torch.device("cuda", 0)
    dt, enq, loss, tr = bench.train_steps("yolov8s-p2-repvgg.yaml", 64, "bf16", steps, 6, 0, 1, dev)
    rec = {"label_path": tag, "ms_per_step": round(dt / steps * 1e3, 3), **tr.bench_gpu}
    rec.pop("gpu_ms_by_step", None)
    print(json.dumps(rec), flush=True)
    del tr
    torch.cuda.empty_cache()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    numpy_table = L.v8DetectionLoss.preprocess
    for rep in range(3):
        L.v8DetectionLoss.preprocess = staticmethod(torch_label_table)
        run("torch-cpu ops (r04)", steps)
        L.v8DetectionLoss.preprocess = staticmethod(numpy_table)
        run("numpy (r05)", steps)
