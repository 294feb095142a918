"""GPU box: what the vendor GEMM (torch.mm -> hipBLASLt) reaches on the 1x1-layer shapes of the B = 256 pass (no bias / activation):
the ceiling a hand-written 1x1 kernel can be compared against.  python tools/gemm_ref_probe.py"""
import torch
dev = torch.device("cuda", 0)
shapes = [("L15 512->256 @40", 409600, 512, 256), ("L17 512->512 @20", 102400, 512, 512), ("L20 768->512 @20", 102400, 768, 512),
          ("L23 1024->512 @20", 102400, 1024, 512), ("L24 768->256 @40", 409600, 768, 256), ("L21 512->256 @20", 102400, 512, 256),
          ("L8 256->128 @80", 1638400, 256, 128), ("L31 192->128 @80", 1638400, 192, 128), ("ref 4096^3", 4096, 4096, 4096), ("ref 8192^3", 8192, 8192, 8192)]
for name, m, k, n in shapes:
    a = (torch.rand(m, k, device=dev) * 2 - 1).half()
    b = (torch.rand(n, k, device=dev) * 2 - 1).half()
    c = torch.empty(m, n, device=dev, dtype=torch.half)
    for _ in range(3):
        torch.mm(a, b.t(), out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    it = 20
    for _ in range(it):
        torch.mm(a, b.t(), out=c)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    print(f"{name:<22s} M={m:<8d} K={k:<5d} N={n:<5d} {us:8.1f} us  {2.0 * m * k * n / us / 1e6:7.1f} TFLOP/s  {(m * k + m * n) * 2 / us / 1e3:6.0f} GB/s", flush=True)
