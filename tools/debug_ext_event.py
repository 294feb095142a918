import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
device = torch.device("cuda", 0)
try:
    ev = torch.cuda.Event(external=True)
    side = torch.cuda.Stream(device=device)
    a = torch.rand(32 * 1024 * 1024, device=device)
    src = torch.zeros(1, device=device); x = torch.zeros(1, device=device); y = torch.zeros(1, device=device)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(40):
            a.mul_(0.999).add_(1e-3)
        x.copy_(src + a[0] * 0.0)
        ev.record()
        for _ in range(8):
            a.mul_(0.999).add_(1e-3)
    print("captured")
    for val in (3.0, 7.0, 11.0):
        src.fill_(val)
        g.replay()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            y.copy_(x)
        torch.cuda.synchronize()
        print("val", val, "y", float(y), "x", float(x))
except Exception:
    traceback.print_exc()
