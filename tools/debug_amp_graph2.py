"""Debug probe 2: the body of tests/test_train_gpu.py::test_fp16_training_runs_under_the_grad_scaler with switches (argv): oracle|nooracle, direct|nodirect"""
import os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.enable()
import torch
from drone_yolo_amd.engine.trainer import DetectionTrainer
sys.path.insert(0, os.path.join(ROOT))
from tests.test_train_gpu import _train_case

sw = set(sys.argv[1:])
device = torch.device("cuda", 0)
g, m, d, model, sd, img, labels = _train_case("tn96")
print("case", m, {k: tuple(v.shape) for k, v in labels.items()}, flush=True)
if "oracle" in sw:
    from oracle import train_oracle as TO
    total_ref, items_ref, grads_ref, _ = TO.loss_and_grads(d, sd, img, labels)
tr = DetectionTrainer(model, dict(optimizer="SGD", lr0=0.001, momentum=0.9, batch=64, dtype="fp16", warmup_epochs=0.0))
batch = dict(img=img.to(device), **labels)
if "direct" in sw:
    loss, _ = tr._forward_backward(batch)
    torch.cuda.synchronize()
    tr.flat.G.zero_()
    if "delloss" in sw:
        del loss, _
    if "fullstep" in sw:
        tr.iters += 1
for it in range(6):
    tr.step(batch, epoch=0, nb=1000)
    torch.cuda.synchronize()
    print(it, tr.amp_state.cpu().tolist(), getattr(tr, "_graph", None) is not None, flush=True)
print("OK", flush=True)
