"""Micro-benchmark of dy_conv2d_nhwc on single layer shapes (GPU box).
usage: python tools/bench_conv.py [--halo 0|1] [--batch B] [shape ...]   shape = cin,cout,k,s,H"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drone_yolo_amd import _lib
from drone_yolo_amd import hip_ops as H

ap = argparse.ArgumentParser()
ap.add_argument("--halo", type=int, default=1)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--lib", default="", help="another build of libdyolo.so, e.g. drone-yolo_amd/lib_ablate/libdyolo.so (make ABLATE=1 OUT=...): only that build reads DYOLO_* probes")
ap.add_argument("--residual", type=int, default=0)
ap.add_argument("shapes", nargs="*", default=["64,64,3,1,160", "32,32,3,1,160", "64,64,3,1,80", "128,128,3,1,40", "256,256,3,1,20",
                                               "512,64,3,1,20", "64,128,3,2,160", "96,64,1,1,160", "768,512,1,1,20", "384,256,1,1,40"])
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp8": H.FP8, "f16x2": H.F16X2}[a.dtype]
if dt == H.FP8:
    H.set_fp8_act_scale(4.0 / 224.0)
dev = torch.device("cuda", 0)
for sh in a.shapes:
    cin, cout, k, s, hh = (int(v) for v in sh.split(","))
    x = torch.randn(a.batch, hh, hh, cin, device=dev)
    if dt == H.F16X2:
        x = H.to_nhwc(x.permute(0, 3, 1, 2).contiguous(), dt)  # split float16: (hi, lo) pairs, 4 bytes per element
    else:
        x = ((x / H.fp8_act_scale()).clamp(-448, 448) if dt == H.FP8 else x).to(dt).permute(0, 3, 1, 2)
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    pc = H.PackedConv(w, torch.zeros(cout), s, k // 2, 1, True, dt, dev, halo=bool(a.halo))
    y = H.conv2d(x, pc)
    res = None
    if a.residual:
        r = torch.randn(a.batch, y.shape[2], y.shape[3], cout, device=dev) * (50.0 if dt == H.FP8 else 1.0)
        res = H.to_nhwc(r.permute(0, 3, 1, 2).contiguous(), dt) if dt == H.F16X2 else r.to(dt).permute(0, 3, 1, 2)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(a.iters):
        H.conv2d(x, pc, out=y, residual=res)
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) / a.iters * 1e3
    fl = 2.0 * a.batch * y.shape[2] * y.shape[3] * cout * cin * k * k
    by = (x.numel() + y.numel()) * (4 if dt == H.F16X2 else x.element_size())
    print(f"{sh:<18s} B={a.batch} {a.dtype} halo={a.halo} {H.last_kernel_name():<34s} dbg={os.environ.get('DYOLO_DBG', '0')}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:7.0f} GB/s(act)")
