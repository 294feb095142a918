"""Where do the small device copies / fills / adds of a training step come from?  (GPU box) torch profiler, grouped by Python stack."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drone_yolo_amd as D
from drone_yolo_amd.engine.trainer import DetectionTrainer
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_state_dict, synthetic_labels

dev = torch.device("cuda", 0)
model = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False)
model.load_state_dict(synthetic_state_dict(model, seed=0))
tr = DetectionTrainer(model, dict(optimizer="SGD", lr0=0.01, momentum=0.937, batch=8, dtype="fp16"))
tr.graph_steps = False
img = torch.randint(0, 256, (8, 3, 640, 640), dtype=torch.uint8).to(dev)
batch = dict(img=img, **synthetic_labels(8, 1000))
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
import collections
acc = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::add_", "aten::add", "aten::zeros", "aten::clone", "aten::mul"):
        st = [s for s in (ev.stack or []) if "drone-yolo_amd" in s or "drone_yolo_amd" in s or "autograd" in s][:2]
        acc[(ev.name, tuple(st))] += 1
for (name, st), n in acc.most_common(40):
    print(n, name, " <- ".join(s.split("/")[-1] for s in st))
