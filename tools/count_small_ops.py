"""GPU box: which Python lines issue the small torch ops (copy_, zeros, fill_, add ...) of one EAGER training step.
usage: DYOLO_TRAIN_GRAPH=0 python tools/count_small_ops.py"""
import collections, os, sys, traceback
os.environ["DYOLO_TRAIN_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from drone_yolo_amd import DetectionModel
from drone_yolo_amd.engine.trainer import DetectionTrainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = collections.Counter()


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func.__name__ if hasattr(func, "__name__") else func)
        if not any(k in name for k in ("view", "detach", "alias", "as_strided", "slice", "select", "permute", "t.default", "expand", "unsqueeze", "reshape", "squeeze", "empty", "size", "stride", "is_", "transpose", "unbind", "split", "_unsafe_view", "narrow", "record_stream", "lift")):
            fr = [f for f in traceback.extract_stack() if ROOT in f.filename and "count_small_ops" not in f.filename]
            where = f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno}" if fr else "?"
            counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda", 0)
B = 8
tr = DetectionTrainer(DetectionModel("yolov8s-p2-repvgg.yaml", nc=10), dict(batch=B, optimizer="SGD", dtype="bf16"))
g = torch.Generator().manual_seed(0)
n = 40
batch = dict(img=torch.randint(0, 255, (B, 3, 640, 640), dtype=torch.uint8, generator=g).to(dev), batch_idx=torch.randint(0, B, (n,), generator=g).float(),
             cls=torch.randint(0, 10, (n, 1), generator=g).float(), bboxes=torch.rand(n, 4, generator=g) * 0.3 + 0.2)
tr.step(batch)
tr.step(batch)
torch.cuda.synchronize()
with torch.autograd.set_multithreading_enabled(False), Count():  # backward on this thread: the mode sees its ops too
    tr.step(batch)
torch.cuda.synchronize()
for (name, where), c in counts.most_common(70):
    print(f"{c:5d}  {name:40s} {where}")
