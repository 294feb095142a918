import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from drone_yolo_amd import hip_ops as H
dt = torch.bfloat16
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
b, cin, cout, h, w = 2, 128, 128, 16, 16
x = torch.randn(b, cin, h, w, generator=g).to(dt).float()
wt = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).to(dt).float()
ref = F.conv2d(x, wt)
pc = H.PackedConv(wt, torch.zeros(cout), 1, 0, 1, False, dt, dev, halo=False)
xd = x.permute(0, 2, 3, 1).contiguous().to(dt).to(dev).permute(0, 3, 1, 2)
y = H.conv2d(xd, pc).float().cpu()
err = (y - ref).abs()
print("max err", float(err.max()))
bad = (err > 0.1).nonzero()
print("bad count", len(bad), "of", err.numel())
if len(bad):
    import collections
    m = (bad[:, 0] * h * w + bad[:, 2] * w + bad[:, 3])
    print("bad pixel idx (flat m) sample", sorted(set(m.tolist()))[:40])
    print("bad couts sample", sorted(set(bad[:, 1].tolist()))[:40])
