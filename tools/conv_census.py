"""Conv census of a model YAML at a given scale / image size: every dense convolution the eval graph runs (RepVGG folded),
its GEMM view and FLOPs.  CPU only (no kernels are launched): python tools/conv_census.py yolov8x-p2-repvgg.yaml 1536"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drone_yolo_amd as D
from drone_yolo_amd.nn.tasks import DetectionModel
from drone_yolo_amd.nn.modules.conv import Conv, PlainConv2d
from drone_yolo_amd.nn.modules.block import RepVGGBlock

cfg, imgsz = sys.argv[1], int(sys.argv[2])
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m = DetectionModel(cfg, nc=nc, verbose=False)
rows = {}
total = 0.0
def hook(name):
    def f(mod, inp, out=None):
        pass
    return f
# walk: compute spatial size per top-level layer from the cumulative stride
for layer in m.model:
    src = m._srcs[layer.i] if hasattr(m, "_srcs") else None
m._plan_graph()
for layer in m.model:
    s_out = m._cum_stride[layer.i]
    srcs = m._srcs[layer.i]
    s_in = 1.0 if layer.i == 0 else m._cum_stride[srcs[0]]
    for name, mod in layer.named_modules():
        if isinstance(mod, Conv):
            c = mod.conv
        elif isinstance(mod, PlainConv2d):
            c = mod
        elif isinstance(mod, RepVGGBlock):
            c = mod.rbr_dense.conv
        else:
            continue
        if type(layer).__name__ == "Detect":
            # per-level: name is cv2.<lvl>.<j> / cv3.<lvl>.<j>
            parts = name.split(".")
            lvl = int(parts[1])
            s = m._cum_stride[srcs[lvl]]
            hin = imgsz / s
        else:
            hin = imgsz / (s_in if (name in ("", "0") or layer.i == 0) else s_out)
            if isinstance(mod, RepVGGBlock) or (isinstance(layer, Conv)):
                hin = imgsz / s_in
        k, st = c.kernel_size[0], c.stride[0]
        hout = hin / st
        fl = 2.0 * c.in_channels * c.out_channels * k * k * hout * hout / max(c.groups, 1)
        total += fl
        key = (c.in_channels, c.out_channels, k, st, int(hin))
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += fl
print(f"{cfg} @{imgsz}: {total/1e9:.1f} GFLOP/img")
for key, (n, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    ci, co, k, st, h = key
    print(f"  {ci:5d}->{co:4d} k{k} s{st} @{h:4d}  x{n:2d}  {fl/1e9:8.1f} GF  {100*fl/total:5.1f}%")
