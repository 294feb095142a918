"""Micro-benchmark of dy_nms on synthetic predictions (GPU box): which phase costs what."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drone_yolo_amd import hip_ops as H

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
B, A, nc = 32, 34000, 10
xy = torch.rand(B, 2, A, generator=g) * 600 + 20
wh = torch.rand(B, 2, A, generator=g) * 40 + 4
sc = torch.rand(B, nc, A, generator=g) ** 6
pred = torch.cat((xy, wh, sc), 1).contiguous().to(dev)
bufs = None
for conf, max_det in ((0.25, 300), (0.25, 10), (0.6, 300), (0.9, 300), (0.05, 300), (0.05, 1)):
    bufs = H.nms(pred, conf, 0.7, max_det=max_det, bufs=None)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        H.nms(pred, conf, 0.7, max_det=max_det, bufs=bufs)
    e.record()
    torch.cuda.synchronize()
    n = (pred[:, 4:].amax(1) > conf).sum(1).float().mean().item()
    print(f"conf {conf} max_det {max_det}: candidates/img {n:.0f} kept {bufs.count.float().mean().item():.0f}  {s.elapsed_time(e) / 10 * 1e3:.1f} us per dy_nms call")
