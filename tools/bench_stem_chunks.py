"""Does running the first two layers (stem 3->32 s2, RepVGG 32->64 s2) per sub-batch keep the 320x320x32 intermediate in
the memory-side cache?  usage (GPU box): python tools/bench_stem_chunks.py [--batch 256]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drone_yolo_amd as D
from drone_yolo_amd import hip_ops as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False).eval().to(dev)
l0, l1 = model.model[0], model.model[1]
B = a.batch
x = torch.rand(B, 3, 640, 640, device=dev)
dt = torch.bfloat16
for sub in (B, 64):
    t = H.alloc_nhwc(sub, 32, 320, 320, dt, dev)
    y = H.alloc_nhwc(B, 64, 160, 160, dt, dev)
    def run():
        for i in range(0, B, sub):
            l0.forward_stem(x[i:i + sub], dt, out=t)
            l1(t, out=y[i:i + sub])
    run(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(a.iters):
        run()
    en.record(); torch.cuda.synchronize()
    print(f"sub-batch {sub:4d}: {st.elapsed_time(en) / a.iters * 1e3:8.1f} us for stem + layer 1 over B={B}  (intermediate {sub * 320 * 320 * 32 * 2 / 1e6:.0f} MB)")

pk = model._stem2_pack(x, dt, consumers0=[1]) if hasattr(model, "_stem2_pack") and (model._plan_graph() or True) else None
if pk is not None:
    y = H.alloc_nhwc(B, 64, 160, 160, dt, dev)
    H.stem2_fused(x, pk, out=y); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(a.iters):
        H.stem2_fused(x, pk, out=y)
    en.record(); torch.cuda.synchronize()
    us = st.elapsed_time(en) / a.iters * 1e3
    print(f"dy_stem2_fused: {us:8.1f} us  ({(x.numel() * 4 + y.numel() * 2) / us / 1e3:.0f} GB/s of image-in + map-out)")
