"""Debug probe 2: which sequence of eager / graphed forward-backward calls makes a replay return a doubled cls loss."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import drone_yolo_amd as D
from drone_yolo_amd.engine.trainer import DetectionTrainer, synthetic_dataset

B = int(sys.argv[1])
seq = sys.argv[2]          # e.g. "ggg", "egg", "geg": e = eager, g = graphed, r = restore BN buffers before the next call
nosink = len(sys.argv) > 3 and sys.argv[3] == "nosink"
device = torch.device("cuda", 0)
data = synthetic_dataset(B, 640, seed=1000)
model = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False)
model.load_state_dict(bench.synthetic_state_dict(model, seed=0))
if nosink:
    DetectionTrainer.grad_sink = False
tr = DetectionTrainer(model, dict(optimizer="SGD", lr0=0.01, momentum=0.937, batch=B, dtype="bf16"))
bn0 = tr.flat.B.clone()
batch = dict(img=data["img"].to(device), batch_idx=data["batch_idx"], cls=data["cls"], bboxes=data["bboxes"])
out = []
for ch in seq:
    if ch == "r":
        tr.flat.B.copy_(bn0)
        continue
    if ch == "d":  # a stream op of the same size that touches nothing the model uses
        dummy = torch.empty_like(bn0)
        dummy.copy_(bn0)
        continue
    if ch == "s":
        torch.cuda.synchronize()
        continue
    if ch == "p":  # perturb the running statistics instead of restoring them
        tr.flat.B.mul_(1.0001)
        continue
    tr.graph_steps = ch == "g"
    tr.iters = 5
    loss, items = tr._forward_backward(batch)
    torch.cuda.synchronize()
    gn = float(tr.flat.G.double().norm())
    tr.flat.G.zero_()
    out.append(f"{ch}:{float(loss):.1f}/cls{float(items[1]):.1f}/gn{gn:.0f}")
    del loss, items
print(B, seq, "nosink" if nosink else "sink", " ".join(out), flush=True)
