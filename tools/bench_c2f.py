"""Time the stride-4 C2f block fused vs layer by layer (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drone_yolo_amd.nn.modules import C2f
from drone_yolo_amd import hip_ops as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
blk = C2f(64, 64, n=1, shortcut=True).eval().to(dev)
x = torch.randn(B, 160, 160, 64, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
for fuse in (True, False):
    blk.fuse_block = fuse
    y = blk(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        blk(x)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    fl = 2.0 * B * 160 * 160 * (64 * 64 + 2 * 32 * 32 * 9 + 96 * 64)
    print(f"fused={fuse}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s (algorithmic)")
