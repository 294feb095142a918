"""Debug probe: graphed vs eager forward/backward of the trainer on the same and on permuted batches (Drone-YOLO-s 640, bf16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import drone_yolo_amd as D
from drone_yolo_amd.engine.trainer import DetectionTrainer, synthetic_dataset

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
device = torch.device("cuda", 0)
data = synthetic_dataset(B, 640, seed=1000)
model = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False)
model.load_state_dict(bench.synthetic_state_dict(model, seed=0))
tr = DetectionTrainer(model, dict(optimizer="SGD", lr0=0.01, momentum=0.937, batch=B, dtype="bf16"))
bn0 = tr.flat.B.clone()

def batch_of(ds, perm=None):
    img, bi, cls, bb = ds["img"], ds["batch_idx"], ds["cls"], ds["bboxes"]
    if perm is not None:
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(len(perm))
        img, bi = img[perm], inv[bi.long()].float()
        order = torch.argsort(bi, stable=True)
        bi, cls, bb = bi[order], cls[order], bb[order]
    return dict(img=img.to(device), batch_idx=bi, cls=cls, bboxes=bb)

def fb(batch, graphed, restore=True):
    if restore:
        tr.flat.B.copy_(bn0)
    tr.graph_steps = graphed
    tr.iters = 5 if graphed else 0
    loss, items = tr._forward_backward(batch)
    torch.cuda.synchronize()
    gn = float(tr.flat.G.double().norm())
    tr.flat.G.zero_()
    return float(loss), [round(float(v), 3) for v in items], gn

perm = torch.randperm(B, generator=torch.Generator().manual_seed(3))
print("eager         ", fb(batch_of(data), False))
print("eager perm    ", fb(batch_of(data, perm), False))
print("graph #1      ", fb(batch_of(data), True))
print("graph #2 same ", fb(batch_of(data), True))
print("graph #3 perm ", fb(batch_of(data, perm), True))
print("graph #4 same ", fb(batch_of(data), True))
print("eager again   ", fb(batch_of(data), False))
print("graph #5 norestore", fb(batch_of(data), True, restore=False))
