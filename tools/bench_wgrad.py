"""Micro-benchmark of dy_conv2d_wgrad_nhwc (GPU box).  usage: python tools/bench_wgrad.py [--batch B] [shape ...]  shape = cin,cout,k,s,H"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drone_yolo_amd import hip_ops as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--eager", action="store_true", help="no hipGraph around the iterations (counter collection)")
ap.add_argument("--lib", default="", help="another build of libdyolo.so (make ABLATE=1 OUT=...): reads DYOLO_WGRAD3_* probes")
ap.add_argument("shapes", nargs="*", default=["32,64,3,2,320", "64,64,3,1,160", "32,32,3,1,160", "64,128,3,2,160", "64,64,3,1,80", "128,128,3,1,40",
                                               "256,256,3,1,20", "128,64,3,1,80", "192,128,1,1,80", "96,64,1,1,160", "768,512,1,1,20"])
a = ap.parse_args()
if a.lib:
    from drone_yolo_amd import _lib

    _lib.LIB_PATH = os.path.abspath(a.lib)
dev = torch.device("cuda", 0)
dt = torch.bfloat16
for sh in a.shapes:
    cin, cout, k, s, hh = (int(v) for v in sh.split(","))
    ho = (hh + 2 * (k // 2) - k) // s + 1
    x = torch.randn(a.batch, hh, hh, cin, device=dev).to(dt).permute(0, 3, 1, 2)
    dz = torch.randn(a.batch, ho, ho, cout, device=dev).to(dt).permute(0, 3, 1, 2)
    buf = torch.zeros(cout, k, k, cin, device=dev)  # the trainer's form: the kernel adds into a gradient sink
    H.conv_wgrad(x, dz, k, s, k // 2, out=buf)
    torch.cuda.synchronize()
    # the iterations are replayed from a hipGraph: a call costs ~80-100 us of Python + allocator time, more than most of these launches
    def run():
        for _ in range(a.iters):
            H.conv_wgrad(x, dz, k, s, k // 2, out=buf)

    if a.eager:
        replay = run
    else:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        replay = gr.replay
    replay()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    replay()
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) / a.iters * 1e3
    fl = 2.0 * a.batch * ho * ho * cout * cin * k * k
    by = (x.numel() + dz.numel()) * 2
    print(f"{sh:<18s} B={a.batch} dbg={os.environ.get('DYOLO_WGRAD3_DBG', '0')} pf2={os.environ.get('DYOLO_WGRAD3_PF2', '-')}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:7.0f} GB/s (x + dz once)")
