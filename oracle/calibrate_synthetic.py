"""ORACLE tooling — calibrates BatchNorm running statistics for bench.py's synthetic weights.

Random conv weights + arbitrary BN statistics do not give a usable network: SiLU has no stable
variance fixed point, so activations either vanish or explode over ~60 layers and the class scores
become input independent (no NMS candidates, or all of them).  A trained network avoids that because
its BN statistics match its data.  This script reproduces that property without training: it runs the
oracle's unfused forward on a few synthetic images with every BatchNorm in "batch statistics" mode,
stores those statistics as the running mean/var (what one training step with momentum 1 would do),
then picks the class-branch bias that lets ~2 % of the anchors clear conf = 0.25.

Output: bench_data/<model>_nc<nc>_seed<seed>_bn.npz (a few tens of KB: BN stats + the bias), read by
bench.py.  The conv weights themselves are regenerated from the seed on every run.

    python oracle/calibrate_synthetic.py [--model yolov8s-p2-repvgg.yaml] [--seed 0]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (synthetic_state_dict: the seeded recipe shared with the benchmark)
import drone_yolo_amd as D  # noqa: E402
from oracle import drone_yolo_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov8s-p2-repvgg.yaml")
    ap.add_argument("--nc", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--target", type=float, default=0.02)
    ap.add_argument("--gamma", type=float, default=1.0, help="scale of every BatchNorm weight: < 1 keeps SiLU near its linear range, which takes a deep RANDOM "
                                                             "network out of the chaotic regime (rounding noise is no longer amplified layer by layer)")
    ap.add_argument("--variant", default="", help="suffix of the output file name")
    a = ap.parse_args()
    torch.set_num_threads(8)
    model = D.DetectionModel(a.model, nc=a.nc, verbose=False)
    sd = bench.synthetic_state_dict(model, a.seed, cls_bias=0.0, bn_stats=None, gamma_scale=a.gamma)
    x = torch.rand(a.images, 3, a.size, a.size, generator=torch.Generator().manual_seed(12345))

    def bn_calibrating(t, sd_, p):
        mean = t.mean((0, 2, 3))
        var = t.var((0, 2, 3), unbiased=False)
        sd_[p + ".running_mean"] = mean
        sd_[p + ".running_var"] = var
        return F.batch_norm(t, mean, var, sd_[p + ".weight"], sd_[p + ".bias"], False, 0.0, O.BN_EPS)

    orig = O.bn_eval
    O.bn_eval = bn_calibrating
    try:
        with torch.no_grad():
            O.forward(model.yaml, sd, x, fused=False)
    finally:
        O.bn_eval = orig
    with torch.no_grad():
        y, feats = O.forward(model.yaml, sd, x, fused=True)
    logits = torch.logit(y[:, 4:].amax(1).clamp(1e-6, 1 - 1e-6)).flatten()
    bias = round(float(math.log(0.25 / 0.75) - torch.quantile(logits, 1 - a.target)), 3)
    sd = bench.synthetic_state_dict(model, a.seed, cls_bias=bias, bn_stats={k: v for k, v in sd.items() if "running_" in k}, gamma_scale=a.gamma)
    with torch.no_grad():
        y, feats = O.forward(model.yaml, sd, x, fused=True)
    frac = float((y[:, 4:].amax(1) > 0.25).float().mean())
    det = O.non_max_suppression(y, 0.25, 0.7, max_det=300, nc=a.nc)
    print(f"cls_bias {bias}  candidates {frac * 100:.2f} %  kept {[len(r) for r in det]}  "
          f"raw logit std {[round(float(f.std()), 2) for f in feats]}")
    out = {k: v.numpy() for k, v in sd.items() if "running_" in k}
    out["__cls_bias__"] = np.array(bias, dtype=np.float32)
    out["__gamma_scale__"] = np.array(a.gamma, dtype=np.float32)
    os.makedirs(os.path.join(ROOT, "bench_data"), exist_ok=True)
    path = os.path.join(ROOT, "bench_data", f"{os.path.splitext(a.model)[0]}_nc{a.nc}_seed{a.seed}{a.variant}_bn.npz")
    np.savez_compressed(path, **out)
    print("wrote", os.path.relpath(path, ROOT), f"{os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
