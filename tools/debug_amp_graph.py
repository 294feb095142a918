"""Debug probe (GPU box): which ingredient of the fp16 + GradScaler step breaks hipGraph capture.  usage: python tools/debug_amp_graph.py <variant>"""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
import drone_yolo_amd as D
from drone_yolo_amd.engine.trainer import DetectionTrainer, synthetic_dataset
from drone_yolo_amd.utils.parity import seeded_state_dict

variant = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else ""
if pre == "cpubwd":
    a = torch.randn(64, 64, requires_grad=True)
    (a @ a).sum().backward()
dtype = {"fp16_amp": "fp16", "fp16_noamp": "fp16", "bf16_amp": "bf16", "bf16": "bf16", "fp32_amp": "fp32", "fp16_amp_seed1": "fp16"}[variant]
dev = torch.device("cuda", 0)
model = D.DetectionModel("yolov8n-p2-repvgg.yaml", nc=10, verbose=False)
model.load_state_dict(seeded_state_dict(model.state_dict(), 5, cls_bias=-1.6))
tr = DetectionTrainer(model, dict(optimizer="SGD", lr0=0.001, momentum=0.9, batch=64, dtype=dtype, warmup_epochs=0.0))
if variant.endswith("noamp"):
    tr.amp_state = None
elif "amp" in variant and tr.amp_state is None:
    tr.amp_state = torch.tensor([1.0 if variant != "fp16_amp_seed1" else 1.0, 0.0, 0.0, 0.0], device=dev)
if variant == "fp16_amp_seed1":
    tr.amp_state[0] = 1.0
data = synthetic_dataset(3, 96, seed=7)
batch = dict(img=data["img"].to(dev), batch_idx=data["batch_idx"], cls=data["cls"], bboxes=data["bboxes"])
if pre == "direct":
    tr._forward_backward(batch)
    torch.cuda.synchronize()
    tr.flat.G.zero_()
if pre == "clone":
    keep = []
for it in range(5):
    if pre == "clone":
        before = tr.flat.P.clone()
        s0 = float(tr.amp_state[0])
    loss, _ = tr.step(batch)
    torch.cuda.synchronize()
    print(variant, it, float(loss), tr.amp_state.cpu().tolist() if tr.amp_state is not None else None, getattr(tr, "_graph", None) is not None, flush=True)
print(variant, "OK", flush=True)
