"""GPU box: which convolutions of one training step (config 3: B = 64, bf16) left BatchNorm statistics from their epilogue, which kernel ran each.
python tools/train_stats_census.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import drone_yolo_amd as D
from drone_yolo_amd.engine.trainer import DetectionTrainer
from drone_yolo_amd.nn import autograd_ops as A

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
model = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False)
model.load_state_dict(bench.synthetic_state_dict(model, seed=0))
tr = DetectionTrainer(model, dict(optimizer="SGD", lr0=0.01, momentum=0.937, batch=batch, dtype="bf16"))
img = torch.randint(0, 256, (batch, 3, 640, 640), generator=torch.Generator().manual_seed(1000), dtype=torch.uint8).to(dev)
b = dict(img=img, **bench.synthetic_labels(batch, 1000))
A.STATS_LOG[0] = log = []
A.BEHIND_LOG[0] = blog = []
tr.step(b)
torch.cuda.synchronize()
A.STATS_LOG[0] = None
A.BEHIND_LOG[0] = None
tot = miss = 0
for w, s, xs, k, slabs in log:
    n_out = xs[0] * w[0] * (xs[2] // s) * (xs[3] // s) * 2
    tot += n_out
    miss += 0 if slabs else n_out
    print(f"{w[1]:>5d}->{w[0]:<5d} k{w[2]} s{s} @{xs[2]:<4d} {k:<44s} slots {slabs:<5d} z {n_out / 1e6:8.1f} MB")
print(f"{len(log)} convolutions, {sum(1 for l in log if not l[4])} without a statistics epilogue: {miss / 1e6:.0f} of {tot / 1e6:.0f} MB of z read again by bn_reduce")
# backward: layers whose output has one consumer (Bottleneck cv1, Detect's first 3x3s) — did that consumer's input-gradient launch leave the sums?
totb = hit = 0
for c, h, w, k, slots in blog:
    nb = batch * c * h * w * 2
    totb += nb
    hit += nb if slots else 0
    print(f"backward sums of {c:>4d} ch @{h:<4d} from {k:<40s} slots {slots:<5d} z {nb / 1e6:8.1f} MB")
print(f"{len(blog)} single-consumer layers, {sum(1 for l in blog if l[4])} with the sums from the gradient epilogue: {hit / 1e6:.0f} of {totb / 1e6:.0f} MB of z (and as much dy) not read by bn_reduce")
