"""Validation step of the training loop on the device (reference: ultralytics/engine/validator.py ``BaseValidator.__call__`` :109-221
in its training branch, and models/yolo/detect/val.py ``DetectionValidator``: ``postprocess`` :93-106 = NMS with ``multi_label=True``
at conf 0.001, ``update_metrics`` :126-174, ``_process_batch`` :213-231, ``get_stats`` :181-190).

What runs where: the model pass (EMA weights, eval mode: BatchNorm folded, the trainer's storage type) and the validator's NMS
(``dy_nms`` with ``multi_label``: one candidate per (anchor, class) pair) on the device; matching and AP on the host in numpy, as in
the reference (utils/metrics.py).  The validation loss is the criterion on Detect's raw maps of the same pass (validator.py:186-187).
Tensor datasets only (dataset files / augmentation are out of scope): ``data["val"]`` — a dict in the training set's
layout — or, when the dataset has no split, the training tensors themselves."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .. import hip_ops as H
from ..utils import ops
from ..utils.metrics import DetMetrics, box_iou, match_predictions


class DetectionValidator:
    def __init__(self, args: Optional[dict] = None):
        a = dict(conf=None, iou=0.7, max_det=300, single_cls=False, agnostic_nms=False)
        a.update({k: v for k, v in (args or {}).items() if k in a})
        self.conf = 0.001 if a["conf"] is None else float(a["conf"])  # default 0.001 in val mode (validator.py:134)
        self.iou, self.max_det = float(a["iou"]), int(a["max_det"])
        self.agnostic = bool(a["single_cls"] or a["agnostic_nms"])
        self.single_cls = bool(a["single_cls"])
        self.iouv = np.linspace(0.5, 0.95, 10)  # detect/val.py:37
        self.metrics = DetMetrics()

    def __call__(self, model, loader, device, dtype: torch.dtype) -> Dict[str, float]:
        """``model``: the DetectionModel carrying the weights to evaluate (the trainer loads the EMA copy), put in eval mode here and
        left there; ``loader``: batches in the training layout (uint8 images + normalised xywh labels).  Returns the reference's
        ``results_dict`` + ``val/*_loss`` + ``fitness``, rounded to 5 digits (validator.py:204-207)."""
        model.eval()
        nc = model.yaml["nc"]
        stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[])
        loss = torch.zeros(3, device=device)
        if getattr(model, "criterion", None) is None:
            model.criterion = model.init_criterion()
        det = model.model[-1]
        fuse_tail = getattr(det, "fuse_tail", False)
        try:
            return self._evaluate(model, loader, device, dtype, nc, stats, loss, det)
        finally:
            det.fuse_tail = fuse_tail  # (a predictor that shares the model keeps its fused tail: ADVICE r4)

    def _evaluate(self, model, loader, device, dtype, nc, stats, loss, det) -> Dict[str, float]:
        nb = 0
        for batch in loader:
            nb += 1
            img = batch["img"].to(device)
            b, _, h, w = img.shape
            x = (img.float() / 255.0) if img.dtype == torch.uint8 else img.float()  # detect/val.py:49-52
            with torch.no_grad():
                det.fuse_tail = False  # the raw maps are needed for the loss: Detect returns (y, feats)
                y, feats = model._predict_once(x.contiguous(), image_dtype=dtype)
                loss += model.criterion(feats, batch)[1]  # validator.py:186-187: loss items of the same pass
                preds = ops.non_max_suppression(y, self.conf, self.iou, nc=nc, multi_label=True, agnostic=self.agnostic, max_det=self.max_det)
            bi = batch["batch_idx"].long().cpu()
            for si, pred in enumerate(preds):  # detect/val.py:126-174
                sel = bi == si
                cls = batch["cls"].cpu()[sel].view(-1).numpy()
                bb = batch["bboxes"].cpu()[sel].float()  # (a dataset held on the device: the scaling below is host arithmetic)
                tbox = (ops.xywh2xyxy(bb) * torch.tensor((w, h, w, h), dtype=torch.float32)).numpy() if len(cls) else np.zeros((0, 4), np.float32)
                pn = pred.cpu().numpy()
                # _prepare_batch / _prepare_pred (detect/val.py:108-124): labels and predictions go through ops.scale_boxes to the original
                # image — for a tensor dataset that is gain 1, pad 0 and the clip to the image (ops.py:92-127, 335-354)
                for arr in (pn, tbox):
                    arr[:, [0, 2]] = arr[:, [0, 2]].clip(0, w)
                    arr[:, [1, 3]] = arr[:, [1, 3]].clip(0, h)
                if self.single_cls:
                    pn[:, 5] = 0
                tp = np.zeros((len(pn), len(self.iouv)), dtype=bool)
                if len(pn) == 0:
                    if len(cls):
                        stats["tp"].append(tp), stats["conf"].append(np.zeros(0)), stats["pred_cls"].append(np.zeros(0)), stats["target_cls"].append(cls)
                    continue
                if len(cls):
                    tp = match_predictions(pn[:, 5], cls, box_iou(tbox, pn[:, :4]), self.iouv)
                stats["tp"].append(tp), stats["conf"].append(pn[:, 4]), stats["pred_cls"].append(pn[:, 5]), stats["target_cls"].append(cls)
        out = {k: 0.0 for k in DetMetrics.keys + ("fitness",)}
        if stats["tp"]:
            cat = {k: np.concatenate(v, 0) for k, v in stats.items()}
            if len(cat["target_cls"]):
                self.metrics.process(cat["tp"], cat["conf"], cat["pred_cls"], cat["target_cls"])
                out = self.metrics.results_dict
        vl = (loss / max(nb, 1)).cpu().tolist()
        out.update({"val/box_loss": vl[0], "val/cls_loss": vl[1], "val/dfl_loss": vl[2]})
        return {k: round(float(v), 5) for k, v in out.items()}
