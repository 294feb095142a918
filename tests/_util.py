"""Helpers shared by the tests (not collected)."""
import ast
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "drone-yolo_amd", "cfg", "models", "v8")


def load_yaml(name, scale, nc):
    d = yaml.safe_load(open(os.path.join(CFG, name)))
    d["scale"], d["nc"] = scale, nc
    return d


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)


def meta(npz, tag):
    return ast.literal_eval(str(npz[f"{tag}__meta"]))


def box_iou_pairs(a, b):
    """IoU of matching rows of two (n,4) xyxy arrays."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    x1, y1 = np.maximum(a[:, 0], b[:, 0]), np.maximum(a[:, 1], b[:, 1])
    x2, y2 = np.minimum(a[:, 2], b[:, 2]), np.minimum(a[:, 3], b[:, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    ua = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter
    return inter / np.maximum(ua, 1e-12)


def split_rows(rows, counts):
    out, o = [], 0
    for c in counts:
        out.append(rows[o : o + int(c)])
        o += int(c)
    return out


def quantize(t, dtype):
    """Round-trip through the device dtype so CPU references see exactly the device's inputs."""
    return t.to(dtype).to(torch.float32)
