import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from drone_yolo_amd import hip_ops as H
dev = torch.device("cuda", 0)
torch.set_printoptions(precision=2, linewidth=250)
dtype = torch.float32
cin, cout = 32, 16
g = torch.Generator().manual_seed(1)
for mode in ("randx_structw", "structx_randw"):
    if mode == "randx_structw":
        x = torch.randn(1, cin, 4, 8, generator=g)
        w = torch.zeros(cout, cin, 1, 1)
        for co in range(cout):
            w[co, co, 0, 0] = 1.0     # y[co] = x[co]
    else:
        x = torch.zeros(1, cin, 4, 8); x[0, 3] = 1.0   # y[co] = w[co][3] at every pixel
        w = torch.randn(cout, cin, 1, 1, generator=g)
    b = torch.zeros(cout)
    pc = H.PackedConv(w, b, 1, 0, 1, False, dtype, dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype).permute(0, 3, 1, 2)
    y = H.conv2d(xd, pc); torch.cuda.synchronize()
    ref = F.conv2d(x, w, b)
    err = (y.cpu() - ref).abs()
    print(mode, "err per pixel (max over ch):", err.amax(1).flatten())
    print(mode, "err per channel:", err.amax((0, 2, 3)))
    p = int(err.amax(1).flatten().argmax())
    print(mode, "worst pixel", p, "got", y.cpu()[0, :, p // 8, p % 8], "ref", ref[0, :, p // 8, p % 8])
