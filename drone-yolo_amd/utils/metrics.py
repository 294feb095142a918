"""Detection metrics of the validation step (reference: ultralytics/utils/metrics.py ``box_iou`` :52-71, ``smooth`` :447-452,
``compute_ap`` :505-534, ``ap_per_class`` :537-623, ``Metric`` :626-760; ``BaseValidator.match_predictions``, engine/validator.py:224-264).

Host logic in numpy, as in the reference (its metrics leave the device too: validator.get_stats() does ``.cpu().numpy()``): a few
thousand kept rows per validation set.  Every function restates the reference's arithmetic step by step and is pinned against the
reference's own functions on synthetic statistics (tests/golden/val_metrics.npz, oracle/make_golden.py::val_metric_vectors)."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz


def box_iou(box1: np.ndarray, box2: np.ndarray, eps: float = 1e-7) -> np.ndarray:
    """(N, 4) x (M, 4) xyxy -> (N, M) IoU in fp32 — metrics.py:52-71."""
    a = box1.astype(np.float32)[:, None, :]
    b = box2.astype(np.float32)[None, :, :]
    inter = np.clip(np.minimum(a[..., 2:], b[..., 2:]) - np.maximum(a[..., :2], b[..., :2]), 0, None).prod(2)
    return inter / ((a[..., 2:] - a[..., :2]).prod(2) + (b[..., 2:] - b[..., :2]).prod(2) - inter + np.float32(eps))


def match_predictions(pred_classes: np.ndarray, true_classes: np.ndarray, iou: np.ndarray, iouv: np.ndarray) -> np.ndarray:
    """(D,) predicted classes, (L,) target classes, (L, D) IoU -> (D, len(iouv)) bool "correct" — validator.py:224-264 (the
    non-scipy branch: per threshold, matches sorted by IoU, each detection and then each label kept once)."""
    correct = np.zeros((pred_classes.shape[0], iouv.shape[0]), dtype=bool)
    correct_class = true_classes[:, None] == pred_classes
    iou = iou * correct_class
    for i, threshold in enumerate(iouv.tolist()):
        matches = np.array(np.nonzero(iou >= threshold)).T
        if matches.shape[0]:
            if matches.shape[0] > 1:
                matches = matches[iou[matches[:, 0], matches[:, 1]].argsort()[::-1]]
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
            correct[matches[:, 1].astype(int), i] = True
    return correct


def smooth(y: np.ndarray, f: float = 0.05) -> np.ndarray:
    """Box filter of fraction f — metrics.py:447-452."""
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def compute_ap(recall: np.ndarray, precision: np.ndarray) -> Tuple[float, np.ndarray, np.ndarray]:
    """101-point interpolated AP of one precision-recall curve — metrics.py:505-534."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    ap = _trapz(np.interp(x, mrec, mpre), x)
    return ap, mpre, mrec


def ap_per_class(tp: np.ndarray, conf: np.ndarray, pred_cls: np.ndarray, target_cls: np.ndarray, eps: float = 1e-16):
    """(tp, fp, p, r, f1, ap (nc, 10), unique_classes) — metrics.py:537-623 without the plots."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes, nt = np.unique(target_cls, return_counts=True)
    nc = unique_classes.shape[0]
    x = np.linspace(0, 1, 1000)
    ap, p_curve, r_curve = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l, n_p = nt[ci], i.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[i]).cumsum(0)
        tpc = tp[i].cumsum(0)
        recall = tpc / (n_l + eps)
        r_curve[ci] = np.interp(-x, -conf[i], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p_curve[ci] = np.interp(-x, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    i = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, i], r_curve[:, i], f1_curve[:, i]
    tp_n = (r * nt).round()
    fp_n = (tp_n / (p + eps) - tp_n).round()
    return tp_n, fp_n, p, r, f1, ap, unique_classes.astype(int)


class DetMetrics:
    """mean precision / recall / mAP50 / mAP50-95 and the fitness the trainer ranks checkpoints by — ``Metric`` (metrics.py:626-760:
    ``mean_results``, ``fitness`` with weights [0, 0, 0.1, 0.9]) behind ``DetMetrics.results_dict`` (:806-880)."""

    keys = ("metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)")

    def __init__(self):
        self.p = self.r = self.f1 = np.zeros(0)
        self.all_ap = np.zeros((0, 10))
        self.ap_class_index = np.zeros(0, dtype=int)

    def process(self, tp, conf, pred_cls, target_cls) -> None:
        res = ap_per_class(tp, conf, pred_cls, target_cls)
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index = res[2], res[3], res[4], res[5], res[6]

    def mean_results(self):
        mp = self.p.mean() if len(self.p) else 0.0
        mr = self.r.mean() if len(self.r) else 0.0
        map50 = self.all_ap[:, 0].mean() if len(self.all_ap) else 0.0
        map_ = self.all_ap.mean() if len(self.all_ap) else 0.0
        return [mp, mr, map50, map_]

    @property
    def fitness(self) -> float:
        return float((np.array(self.mean_results()) * np.array([0.0, 0.0, 0.1, 0.9])).sum())

    @property
    def results_dict(self) -> Dict[str, float]:
        return dict(zip(self.keys + ("fitness",), [float(v) for v in self.mean_results()] + [self.fitness]))
