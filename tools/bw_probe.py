"""Practical HBM ceiling on this box: device-to-device copy and read-only reduction bandwidth (torch kernels)."""
import torch, time
dev = torch.device("cuda", 0)
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        b.copy_(a)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print(f"copy {mb} MB: {2 * n * 2 / t / 1e12:.2f} TB/s (read+write)")
    s.record()
    for _ in range(20):
        a.float().sum() if False else torch.sum(a, dtype=torch.float32)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print(f"sum  {mb} MB: {n * 2 / t / 1e12:.2f} TB/s (read only)")
