"""Parity statistics of a detection pass against the reference's CPU fp32 result ("parity gate reported
next to every throughput number": class / index exact-match rate and min box IoU).

The expectation comes from ``tests/golden/*.npz`` — outputs of the REAL reference (`oracle/make_golden.py` imports
/root/reference in the build container and stores its ``non_max_suppression`` rows and kept anchor indices); this module
only compares numbers and holds no model code, so ``bench.py`` can print the gate without touching ``oracle/``.
"""
from __future__ import annotations

import ast
import math
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def seeded_state_dict(template: Dict[str, torch.Tensor], seed: int, cls_bias: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """The name-ordered seeded weight generator the golden fixtures were made with (fixtures hold outputs only; both sides
    regenerate the weights from the seed): one CPU generator walked over the sorted keys — conv ~ N(0, 2/fan_in), BatchNorm
    affine near identity with non-trivial running statistics, DFL arange, class-branch bias ``cls_bias`` + a small ramp.
    ``tests/test_oracle_golden.py`` checks it is bit-identical to the generator the reference was loaded with."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(template.keys()):
        t = template[k]
        shape = tuple(t.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=t.dtype)
        elif ".dfl." in k or k.startswith("dfl."):
            out[k] = torch.arange(shape[1], dtype=torch.float32).view(shape)
        elif k.endswith("running_mean"):
            out[k] = torch.randn(shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            out[k] = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif k.endswith("weight") and len(shape) == 1:
            out[k] = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif k.endswith("bias"):
            out[k] = torch.randn(shape, generator=g) * 0.1
        else:
            out[k] = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
    if cls_bias is not None:
        for k in out:
            if ".cv3." in k and k.endswith(".2.bias"):
                out[k] = out[k] * 0 + cls_bias + torch.linspace(-0.3, 0.3, out[k].numel())
    return out


def box_iou_pairs(a, b) -> np.ndarray:
    """IoU of matching rows of two (n, 4) xyxy arrays (float64)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    x1, y1 = np.maximum(a[:, 0], b[:, 0]), np.maximum(a[:, 1], b[:, 1])
    x2, y2 = np.minimum(a[:, 2], b[:, 2]), np.minimum(a[:, 3], b[:, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    ua = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter
    return inter / np.maximum(ua, 1e-12)


def match_stats(rows, idx, exp_rows, exp_idx) -> Tuple[float, float, float]:
    """One image: (fraction of the reference detections reproduced with the same anchor index AND class, min IoU and mean
    IoU of those)."""
    exp_map = {int(a): r for a, r in zip(exp_idx, exp_rows)}
    got_map = {int(a): r for a, r in zip(idx, rows)}
    if not exp_map:
        return (1.0 if not got_map else 0.0), 1.0, 1.0
    common = [a for a in exp_map if a in got_map and int(got_map[a][5]) == int(exp_map[a][5])]
    if not common:
        return 0.0, 0.0, 0.0
    ious = box_iou_pairs(np.stack([got_map[a][:4] for a in common]), np.stack([exp_map[a][:4] for a in common]))
    return len(common) / len(exp_map), float(ious.min()), float(ious.mean())


def split_rows(rows, counts):
    out, o = [], 0
    for c in counts:
        out.append(rows[o : o + int(c)])
        o += int(c)
    return out


def clip_rows(rows: np.ndarray, hw: Sequence[int]) -> np.ndarray:
    """construct_result's clip of the kept boxes to the image (detect/predict.py:59-73, ops.py:335-354)."""
    r = rows.copy()
    if len(r):
        r[:, [0, 2]] = r[:, [0, 2]].clip(0, hw[1])
        r[:, [1, 3]] = r[:, [1, 3]].clip(0, hw[0])
    return r


def golden_case(npz_name: str, tag: str):
    """(meta dict, input tensor, expected rows per image, expected kept anchor indices per image) of one end-to-end fixture."""
    g = np.load(os.path.join(ROOT, "tests", "golden", npz_name), allow_pickle=False)
    meta = ast.literal_eval(str(g[f"{tag}__meta"]))
    b, h, w = meta["shape"]
    if "seed" in meta:
        x = torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(meta["seed"]))
    else:
        x = None  # the caller builds it (tiles of a frame)
    exp_rows = [clip_rows(r, (h, w)) for r in split_rows(g[f"{tag}__det"], g[f"{tag}__n"])]
    exp_idx = split_rows(g[f"{tag}__det_idx"], g[f"{tag}__n"])
    return meta, x, exp_rows, exp_idx


def detection_parity(nms_bufs, exp_rows, exp_idx, conf: float = 0.25, margin: float = 0.0) -> dict:
    """Gate numbers of one pass (``nms_bufs``: hip_ops.NmsBuffers of the device pass) against the reference rows.  Two-sided:
    ``missed`` = reference detections this pass does not keep with the same (anchor index, class); ``extra`` = detections this pass
    keeps that the reference does not (same key) — so a pass that kept everything would not read as a perfect match.
    ``missed_clear`` / ``extra_clear`` count only those scored beyond ``conf + margin`` (reference score for a missed one, this pass's
    for an extra one): a detection scored within the storage type's score error of the confidence threshold is decided by rounding,
    whatever the kernels do — the clear ones are NMS near-ties or real defects."""
    counts = nms_bufs.count.cpu().tolist()
    out_rows, out_idx = nms_bufs.out.cpu().numpy(), nms_bufs.index.cpu().numpy()
    stats, sets_equal, n_ref, n_hit, n_got, missed_img, extra_img, missed_clear, extra_clear = [], True, 0, 0, 0, [], [], 0, 0
    for i, c in enumerate(counts):
        st = match_stats(out_rows[i, :c], out_idx[i, :c], exp_rows[i], exp_idx[i])
        stats.append(st)
        ref_keys = {(int(a), int(r[5])): float(r[4]) for a, r in zip(exp_idx[i], exp_rows[i])}
        got_keys = {(int(a), int(r[5])): float(r[4]) for a, r in zip(out_idx[i, :c], out_rows[i, :c])}
        n_ref += len(ref_keys)
        n_got += len(got_keys)
        n_hit += len(ref_keys.keys() & got_keys.keys())
        mi = [k for k in ref_keys if k not in got_keys]
        ex = [k for k in got_keys if k not in ref_keys]
        missed_img.append(len(mi))
        extra_img.append(len(ex))
        missed_clear += sum(1 for k in mi if ref_keys[k] > conf + margin)
        extra_clear += sum(1 for k in ex if got_keys[k] > conf + margin)
        sets_equal &= sorted(out_idx[i, :c].tolist()) == sorted(int(a) for a in exp_idx[i])
    missed, extra = sum(missed_img), sum(extra_img)
    return {"images": len(counts), "ref_detections": n_ref, "kept_detections": n_got, "match_rate": round(n_hit / max(n_ref, 1), 5),
            "missed": missed, "extra": extra, "missed_frac": round(missed / max(n_ref, 1), 5), "extra_frac": round(extra / max(n_ref, 1), 5),
            "missed_clear": missed_clear, "extra_clear": extra_clear, "clear_margin": margin,
            "missed_max_image": max(missed_img, default=0), "extra_max_image": max(extra_img, default=0),
            "match_rate_min_image": round(min(s[0] for s in stats), 5), "iou_min": round(min(s[1] for s in stats), 6),
            "iou_mean": round(float(np.mean([s[2] for s in stats])), 6), "counts_equal": counts == [len(r) for r in exp_idx],
            "kept_sets_identical": bool(sets_equal)}
