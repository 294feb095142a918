"""Micro-benchmark of dy_bn_train_fwd / dy_bn_train_bwd (GPU box).  usage: python tools/bench_bn.py [--batch B] [shape ...]  shape = c,H
Buffers rotate through a ring larger than the 256 MB memory-side cache, so every pass streams from HBM as it does inside a training step."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drone_yolo_amd import hip_ops as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--ring_mb", type=int, default=1024)
ap.add_argument("--stats", type=int, default=0, help="> 0: forward as after a convolution with a statistics epilogue (that many slots already in the workspace)")
ap.add_argument("--lib", default="", help="another build of libdyolo.so (make ABLATE=1 OUT=...): reads the DYOLO_BN_* probes")
ap.add_argument("shapes", nargs="*", default=["32,160", "64,160", "64,80", "128,80", "128,40", "256,40", "256,20", "512,20"])
a = ap.parse_args()
if a.lib:
    from drone_yolo_amd import _lib

    _lib.LIB_PATH = os.path.abspath(a.lib)
dev = torch.device("cuda", 0)
dt = torch.bfloat16
tot = {"fwd": 0.0, "bwd": 0.0}
for sh in a.shapes:
    c, hh = (int(v) for v in sh.split(","))
    one = a.batch * hh * hh * c * 2
    nring = max(2, min(64, (a.ring_mb << 20) // (3 * one)))
    zs = [torch.randn(a.batch, hh, hh, c, device=dev).to(dt).permute(0, 3, 1, 2) for _ in range(nring)]
    dys = [torch.randn(a.batch, hh, hh, c, device=dev).to(dt).permute(0, 3, 1, 2) for _ in range(nring)]
    outs = [H.alloc_nhwc(a.batch, c, hh, hh, dt, dev) for _ in range(nring)]
    g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    st = H.BnState(c, dev)
    H.bn_train_fwd(zs[0], g, b, st, True, running_mean=rm, running_var=rv, out=outs[0])
    torch.cuda.synchronize()
    iters = max(nring, 8)

    def fwd():
        for i in range(iters):
            H.bn_train_fwd(zs[i % nring], g, b, st, True, running_mean=rm, running_var=rv, out=outs[i % nring], partial_slabs=a.stats)

    def bwd():
        for i in range(iters):
            H.bn_train_bwd(dys[i % nring], zs[i % nring], g, b, st, True, out=outs[i % nring])

    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters * 1e3
    nf = 2 if a.stats else 3  # tensor passes: forward z (stats) + z + y; backward (dy + z) twice + dz
    tot["fwd"] += res["fwd"]
    tot["bwd"] += res["bwd"]
    print(f"c={c:<4d} {hh}x{hh} B={a.batch} ring={nring}: fwd {res['fwd']:7.1f} us {nf * one / res['fwd'] / 1e3:6.0f} GB/s   bwd {res['bwd']:7.1f} us {5 * one / res['bwd'] / 1e3:6.0f} GB/s", flush=True)
print(f"sum fwd {tot['fwd']:.1f} us  bwd {tot['bwd']:.1f} us")
