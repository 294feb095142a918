"""Micro-benchmark of dy_c2f_fused (GPU box): the backbone block (64 -> 64) and the neck block (upsampled 128 + 64 -> 64).
usage: python tools/bench_c2f.py [--batch B] [--size S] [--lib other/libdyolo.so]   (an ABLATE build reads DYOLO_C2F_DBG)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drone_yolo_amd import _lib
from drone_yolo_amd import hip_ops as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--size", type=int, default=160)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--lib", default="")
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for c_lo in (0, 128):
    cin = 64 + c_lo
    mk = lambda co, ci, k: (torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5, torch.randn(co, generator=g) * 0.1)  # noqa: E731
    pk = H.PackedC2f(mk(64, cin, 1), mk(32, 32, 3), mk(32, 32, 3), mk(64, 96, 1), shortcut=(c_lo == 0), dtype=dt, device=dev)
    x = torch.randn(a.batch, a.size, a.size, 64, device=dev).to(dt).permute(0, 3, 1, 2)
    x_lo = torch.randn(a.batch, a.size // 2, a.size // 2, c_lo, device=dev).to(dt).permute(0, 3, 1, 2) if c_lo else None
    y = H.c2f_fused(x, pk, x_lo=x_lo)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(a.iters):
        H.c2f_fused(x, pk, out=y, x_lo=x_lo)
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) / a.iters * 1e3
    px = a.batch * a.size * a.size
    fl = 2.0 * px * (cin * 64 + 2 * 9 * 32 * 32 + 96 * 64)
    by = (x.numel() + y.numel() + (x_lo.numel() if c_lo else 0)) * 2
    if a.lib:
        import ctypes
        h = ctypes.CDLL(os.path.abspath(a.lib))
        if hasattr(h, "dy_c2f_debug_phase_cycles"):
            buf = (ctypes.c_ulonglong * 8)()
            h.dy_c2f_debug_phase_cycles(buf, 1)
            tot = sum(buf[:4]) or 1
            print("   phase share of block 0 / wave 0 (A, B, C, D+E):", [round(buf[k] / tot, 3) for k in range(4)], "cycles/strip (100 MHz clock ticks x?)", [buf[k] // ((a.iters + 1) * 25) for k in range(4)])
    print(f"C2f cin {cin} (upsampled {c_lo}) B={a.batch} {a.size}x{a.size} dbg={os.environ.get('DYOLO_C2F_DBG', '0')}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:7.0f} GB/s")
