"""Detection training loss on the device (reference: ultralytics/utils/loss.py:157-260 ``v8DetectionLoss``).

TaskAlignedAssigner + BCE / CIoU / DFL, value and gradient, through ``dy_detection_loss``.  When the head maps carry an
autograd graph (training forward) the returned loss is differentiable: ``loss.backward()`` feeds the kernel's gradient
w.r.t. the head outputs into the backward of the model (nn/autograd_ops.py)."""
from __future__ import annotations

import torch

from .. import hip_ops as H
from .ops import xywh2xyxy


class v8DetectionLoss:
    """Same constructor idea and call contract as the reference: ``loss, loss_items = criterion(preds, batch)``."""

    def __init__(self, model, tal_topk: int = 10, box: float = 7.5, cls: float = 0.5, dfl: float = 1.5):
        m = model.model[-1]  # Detect()
        h = getattr(model, "args", None)
        self.box, self.cls, self.dfl = (getattr(h, "box", box), getattr(h, "cls", cls), getattr(h, "dfl", dfl)) if h else (box, cls, dfl)
        self.stride, self.nc, self.reg_max, self.no = m.stride, m.nc, m.reg_max, m.nc + m.reg_max * 4
        self.topk = tal_topk

    @staticmethod
    def preprocess(targets: torch.Tensor, batch_size: int, scale_tensor: torch.Tensor) -> torch.Tensor:
        """(N, 6) [image, cls, x, y, w, h normalised] -> (B, n_max, 5) [cls, x1, y1, x2, y2] pixels — loss.py:180-195.
        Vectorised on the host (labels arrive on the host from the loader; one scatter instead of the reference's per-image loop,
        no device synchronisation)."""
        targets = targets.detach().cpu().float()
        nl, ne = targets.shape
        if nl == 0:
            return torch.zeros(batch_size, 0, ne - 1)
        img = targets[:, 0].long()
        counts = torch.bincount(img, minlength=batch_size)
        order = torch.argsort(img, stable=True)
        start = torch.cumsum(counts, 0) - counts
        pos = torch.arange(nl) - start[img[order]]  # rank of each label inside its image, in the order the labels came
        out = torch.zeros(batch_size, int(counts.max()), ne - 1)
        out[img[order], pos] = targets[order, 1:]
        out[..., 1:5] = xywh2xyxy(out[..., 1:5] * scale_tensor)
        return out

    def targets_to_gt(self, batch, batch_size: int, img_hw) -> torch.Tensor:
        """The batch's labels as the (B, n_max, 5) pixel-box table on the host (``preprocess``); img_hw = (h, w) of the images."""
        imgsz = torch.tensor([float(img_hw[0]), float(img_hw[1])], dtype=torch.float32)
        targets = torch.cat((batch["batch_idx"].view(-1, 1).float().cpu(), batch["cls"].view(-1, 1).float().cpu(),
                             batch["bboxes"].float().cpu()), 1)
        return self.preprocess(targets, batch_size, imgsz[[1, 0, 1, 0]])

    def __call__(self, preds, batch):
        feats = preds[1] if isinstance(preds, tuple) else preds
        hw = (feats[0].shape[2] * float(self.stride[0]), feats[0].shape[3] * float(self.stride[0]))
        return self.from_gt(preds, self.targets_to_gt(batch, feats[0].shape[0], hw))

    def from_gt(self, preds, gt: torch.Tensor):
        """loss, loss_items from the box table itself (host or device, zero rows = padding): what ``__call__`` does after
        ``preprocess``.  A trainer that replays the step as a hipGraph keeps ``gt`` in a static device buffer."""
        feats = preds[1] if isinstance(preds, tuple) else preds
        strides = [float(s) for s in self.stride]
        if any(f.requires_grad for f in feats):
            if self.topk != 10:
                raise NotImplementedError("the differentiable path uses the default tal_topk of 10")
            from ..nn.autograd_ops import DetectionLossFn

            total, items = DetectionLossFn.apply(gt, strides, self.nc, self.reg_max, (self.box, self.cls, self.dfl), *feats)
            return total, items
        out, _ = H.detection_loss(feats, gt, strides, self.nc, self.reg_max, topk=self.topk, box=self.box, cls=self.cls, dfl=self.dfl)
        return out[3], out[:3]
