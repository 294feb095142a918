"""Tile study of the split-float16 kernel (ablate build: make ABLATE=1 OUT=../lib_ablate): every DYOLO_SPLIT_CFG on the layer shapes of
Drone-YOLO-s at B = 256.    python tools/split_tiles.py [--batch 256] [shape ...]   shape = cin,cout,k,s,H[,res]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drone_yolo_amd import _lib
from drone_yolo_amd import hip_ops as H

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--cfgs", default="0,1,2,3,4,5,6,7,8,9,10,11,12,13")
ap.add_argument("shapes", nargs="*", default=["32,32,3,1,160", "64,64,3,1,160", "64,64,3,1,80", "64,64,1,1,160", "96,64,1,1,160", "128,64,3,1,80", "256,64,3,1,40",
                                               "128,128,3,1,40", "256,256,3,1,20", "64,128,3,2,160", "192,128,1,1,80", "384,256,1,1,40", "768,512,1,1,20", "32,64,3,2,320"])
a = ap.parse_args()
_lib.LIB_PATH = os.path.abspath("drone-yolo_amd/lib_ablate/libdyolo.so")
dev = torch.device("cuda", 0)
for sh in a.shapes:
    cin, cout, k, s, hh = (int(v) for v in sh.split(",")[:5])
    x = H.to_nhwc(torch.randn(a.batch, cin, hh, hh, device=dev), H.F16X2)
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    pc = H.PackedConv(w, torch.zeros(cout), s, k // 2, 1, True, H.F16X2, dev)
    row = []
    for cfg in (int(c) for c in a.cfgs.split(",")):
        bn = {0: 0, 1: 64, 2: 64, 3: 64, 4: 64, 5: 32, 6: 32, 7: 32, 8: 128, 9: 128, 10: 128, 11: 128, 12: 64, 13: 32}[cfg]
        words = k * k * cin // 4 if k == 3 else 0
        if cfg and (words > (640 if bn == 128 else 256) or (bn < cout and cout % bn) or (bn > 2 * cout)):
            continue
        os.environ["DYOLO_SPLIT_CFG"] = str(cfg)
        try:
            y = H.conv2d(x, pc)
        except Exception as e:  # noqa: BLE001
            print("  cfg", cfg, "refused:", str(e)[:80], flush=True)
            continue
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(a.iters):
            H.conv2d(x, pc, out=y)
        en.record()
        torch.cuda.synchronize()
        us = st.elapsed_time(en) / a.iters * 1e3
        fl = 2.0 * a.batch * y.shape[2] * y.shape[3] * cout * cin * k * k
        row.append((us, cfg, H.last_kernel_name(), fl / us / 1e6))
    best = min(row)
    print(f"{sh:<18s} " + "  ".join(f"[{c}] {u:7.1f}us {t:5.0f}TF" for u, c, _, t in row) + f"   best [{best[1]}] {best[2]}", flush=True)
