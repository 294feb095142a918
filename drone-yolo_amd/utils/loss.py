"""Detection training loss on the device (reference: ultralytics/utils/loss.py:157-260 ``v8DetectionLoss``).

TaskAlignedAssigner + BCE / CIoU / DFL, value and gradient, through ``dy_detection_loss``.  When the head maps carry an
autograd graph (training forward) the returned loss is differentiable: ``loss.backward()`` feeds the kernel's gradient
w.r.t. the head outputs into the backward of the model (nn/autograd_ops.py)."""
from __future__ import annotations

import numpy as np
import torch

from .. import hip_ops as H


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
        no device synchronisation).  numpy, not torch: a torch CPU op above the intra-op grain wakes the whole OpenMP pool, which on
        a GPU box is sized by the HOST's core count while the process owns a 16-core CFS quota — the pool's spin-waits spend the
        quota and the kernel parks every thread of the process until the next 100 ms period (r05: one such stall per 30-60 training
        steps, each longer than the ~3 steps of work the launch queue holds)."""
        t = targets.detach().cpu().float().numpy()
        nl, ne = t.shape
        if nl == 0:
            return torch.zeros(batch_size, 0, ne - 1)
        img = t[:, 0].astype(np.int64)
        counts = np.bincount(img, minlength=batch_size)
        order = np.argsort(img, kind="stable")
        start = np.cumsum(counts) - counts
        pos = np.arange(nl) - start[img[order]]  # rank of each label inside its image, in the order the labels came
        out = np.zeros((batch_size, int(counts.max()), ne - 1), dtype=np.float32)
        out[img[order], pos] = t[order, 1:]
        sc = scale_tensor.detach().cpu().float().numpy()
        xy, half = out[..., 1:3] * sc[:2], out[..., 3:5] * sc[2:] / np.float32(2)  # xywh2xyxy (utils/ops.py:432-449) in fp32, as torch computes it
        out[..., 1:3], out[..., 3:5] = xy - half, xy + half
        return torch.from_numpy(out)

    def targets_to_gt(self, batch, batch_size: int, img_hw) -> torch.Tensor:
        """The batch's labels as the (B, n_max, 5) pixel-box table on the host (``preprocess``); img_hw = (h, w) of the images."""
        imgsz = torch.tensor([float(img_hw[1]), float(img_hw[0]), float(img_hw[1]), float(img_hw[0])], dtype=torch.float32)
        targets = torch.cat((batch["batch_idx"].view(-1, 1).float().cpu(), batch["cls"].view(-1, 1).float().cpu(),
                             batch["bboxes"].float().cpu()), 1)
        return self.preprocess(targets, batch_size, imgsz)

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
