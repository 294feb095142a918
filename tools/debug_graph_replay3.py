"""Debug probe 3: after a good and a bad replay, which BatchNorm layer's running statistics differ first (= first divergent layer of the forward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import drone_yolo_amd as D
from drone_yolo_amd.engine.trainer import DetectionTrainer, synthetic_dataset

B = int(sys.argv[1])
device = torch.device("cuda", 0)
data = synthetic_dataset(B, 640, seed=1000)
model = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False)
model.load_state_dict(bench.synthetic_state_dict(model, seed=0))
tr = DetectionTrainer(model, dict(optimizer="SGD", lr0=0.01, momentum=0.937, batch=B, dtype="bf16"))
bn0 = tr.flat.B.clone()
batch = dict(img=data["img"].to(device), batch_idx=data["batch_idx"], cls=data["cls"], bboxes=data["bboxes"])
names = [k for k, b in model.named_buffers() if b.is_floating_point()]
sizes = [b.numel() for k, b in model.named_buffers() if b.is_floating_point()]
res = []
for rep in range(3):
    tr.flat.B.copy_(bn0)
    tr.graph_steps, tr.iters = True, 5
    loss, items = tr._forward_backward(batch)
    torch.cuda.synchronize()
    res.append((float(loss), tr.flat.B.clone(), tr.flat.G.clone()))
    tr.flat.G.zero_()
    del loss, items
print("losses", [r[0] for r in res])
off = 0
shown = 0
for k, c in zip(names, sizes):
    a, b = res[0][1][off:off + c], res[1][1][off:off + c]
    d = float((a - b).abs().max()) / max(float(a.abs().max()), 1e-9)
    if d > 1e-4 and shown < 12:
        print(f"B diff {k}: rel {d:.3e}  good {a[:3].tolist()} bad {b[:3].tolist()}")
        shown += 1
    off += c
shown = 0
for k, (o, c) in tr.flat.offsets.items():
    a, b = res[0][2][o:o + c], res[1][2][o:o + c]
    d = float((a - b).norm()) / max(float(a.norm()), 1e-12)
    if d > 1e-3 and shown < 400:
        shown += 1
print("params with different gradient:", shown, "of", len(tr.flat.offsets))
