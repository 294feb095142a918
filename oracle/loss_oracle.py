"""ORACLE — test infrastructure, not product code.

CPU restatement (plain PyTorch fp32) of the reference's detection training loss: `bbox_iou` with the CIoU
term (ultralytics/utils/metrics.py:74-134), `TaskAlignedAssigner` (utils/tal.py:14-295), `bbox2dist`
(tal.py:360-363), `DFLoss` / `BboxLoss` (utils/loss.py:65-113) and `v8DetectionLoss` (loss.py:157-260).
Written as explicit per-step tensor code (dense (B, G, A) tensors like the reference — sizes in tests are
small) with every step citing the line it follows.  Pinned against the real reference by
`oracle/make_golden.py` (section `loss_vectors`), fixtures in `tests/golden/loss.npz`.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from oracle.drone_yolo_oracle import dist2bbox, make_anchors, xywh2xyxy

Tensor = torch.Tensor


def bbox_ciou(box1: Tensor, box2: Tensor, eps: float = 1e-7) -> Tensor:
    """CIoU of xyxy boxes, broadcasting on leading dims; returns (..., 1) — metrics.py:74-134 with xywh=False, CIoU=True."""
    b1x1, b1y1, b1x2, b1y2 = box1.chunk(4, -1)
    b2x1, b2y1, b2x2, b2y2 = box2.chunk(4, -1)
    w1, h1 = b1x2 - b1x1, b1y2 - b1y1 + eps  # metrics.py:104-105 (eps on heights only)
    w2, h2 = b2x2 - b2x1, b2y2 - b2y1 + eps
    inter = (torch.minimum(b1x2, b2x2) - torch.maximum(b1x1, b2x1)).clamp(min=0) * (
        torch.minimum(b1y2, b2y2) - torch.maximum(b1y1, b2y1)).clamp(min=0)  # :108-110
    union = w1 * h1 + w2 * h2 - inter + eps  # :113
    iou = inter / union
    cw = torch.maximum(b1x2, b2x2) - torch.minimum(b1x1, b2x1)  # :118
    ch = torch.maximum(b1y2, b2y2) - torch.minimum(b1y1, b2y1)
    c2 = cw.pow(2) + ch.pow(2) + eps  # :121
    rho2 = ((b2x1 + b2x2 - b1x1 - b1x2).pow(2) + (b2y1 + b2y2 - b1y1 - b1y2).pow(2)) / 4  # :122-124
    v = (4 / math.pi**2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)).pow(2)  # :126
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))  # :127-128
    return iou - (rho2 / c2 + v * alpha)  # :129


def bbox2dist(anchor_points: Tensor, bbox: Tensor, reg_max: float) -> Tensor:
    """xyxy -> (l, t, r, b) distances clamped to [0, reg_max - 0.01] — tal.py:360-363."""
    x1y1, x2y2 = bbox.chunk(2, -1)
    return torch.cat((anchor_points - x1y1, x2y2 - anchor_points), -1).clamp(0, reg_max - 0.01)


def task_aligned_assign(pd_scores: Tensor, pd_bboxes: Tensor, anc_points: Tensor, gt_labels: Tensor, gt_bboxes: Tensor,
                        mask_gt: Tensor, topk: int = 10, num_classes: int = 80, alpha: float = 0.5, beta: float = 6.0,
                        eps: float = 1e-9):
    """TaskAlignedAssigner.forward — tal.py:39-118.  Shapes: pd_scores (B,A,nc) probabilities, pd_bboxes (B,A,4) xyxy
    pixels, anc_points (A,2) pixels, gt_labels (B,G,1), gt_bboxes (B,G,4), mask_gt (B,G,1).
    Returns target_labels (B,A), target_bboxes (B,A,4), target_scores (B,A,nc), fg_mask (B,A) bool, target_gt_idx (B,A)."""
    B, A, _ = pd_scores.shape
    G = gt_bboxes.shape[1]
    if G == 0:  # tal.py:64-71
        return (torch.full((B, A), num_classes, dtype=pd_scores.dtype), torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores),
                torch.zeros((B, A), dtype=torch.bool), torch.zeros((B, A), dtype=torch.int64))
    # anchors strictly inside each gt box: min(l,t,r,b) > eps — tal.py:241-262
    lt, rb = gt_bboxes.view(-1, 1, 4).chunk(2, 2)
    deltas = torch.cat((anc_points[None] - lt, rb - anc_points[None]), dim=2).view(B, G, A, 4)
    mask_in_gts = (deltas.amin(3) > eps).to(gt_bboxes.dtype)
    # alignment metric s^alpha * ciou^beta on (in-gt & valid-gt) pairs — tal.py:132-155
    valid = (mask_in_gts * mask_gt).bool()  # (B,G,A)
    cls_idx = gt_labels.squeeze(-1).long()  # (B,G)
    scores_gt = pd_scores.permute(0, 2, 1)[torch.arange(B)[:, None], cls_idx]  # (B,G,A): score of the gt's class at every anchor
    bbox_scores = torch.where(valid, scores_gt, torch.zeros_like(scores_gt))
    ciou = bbox_ciou(gt_bboxes[:, :, None, :], pd_bboxes[:, None, :, :]).squeeze(-1).clamp(min=0)  # tal.py:153-155
    overlaps = torch.where(valid, ciou, torch.zeros_like(ciou))
    align_metric = bbox_scores.pow(alpha) * overlaps.pow(beta)
    # top-k anchors per gt (tal.py:157-190); padded gts point all k picks at anchor 0, which the count>1 rule then drops
    topk_idx = torch.topk(align_metric, topk, dim=-1, largest=True).indices  # (B,G,k)
    topk_idx = topk_idx.masked_fill(~mask_gt.expand(-1, -1, topk).bool(), 0)
    count = torch.zeros_like(align_metric, dtype=torch.int32)
    count.scatter_add_(-1, topk_idx, torch.ones_like(topk_idx, dtype=torch.int32))
    mask_topk = torch.where(count > 1, torch.zeros_like(count), count).to(align_metric.dtype)
    mask_pos = mask_topk * mask_in_gts * mask_gt  # tal.py:128
    # an anchor claimed by several gts goes to the gt with the highest overlap (over ALL gts) — tal.py:265-295
    fg = mask_pos.sum(-2)
    if fg.max() > 1:
        multi = (fg.unsqueeze(1) > 1).expand(-1, G, -1)
        best_gt = overlaps.argmax(1)  # (B,A)
        one_hot = torch.zeros_like(mask_pos).scatter_(1, best_gt.unsqueeze(1), 1)
        mask_pos = torch.where(multi, one_hot, mask_pos).float()
        fg = mask_pos.sum(-2)
    target_gt_idx = mask_pos.argmax(-2)  # (B,A)
    # targets — tal.py:192-239
    flat_idx = target_gt_idx + torch.arange(B)[:, None] * G
    target_labels = gt_labels.long().flatten()[flat_idx].clamp(min=0)
    target_bboxes = gt_bboxes.view(-1, 4)[flat_idx]
    target_scores = F.one_hot(target_labels, num_classes).to(torch.int64)
    target_scores = torch.where(fg[:, :, None] > 0, target_scores, torch.zeros_like(target_scores))
    # normalised alignment as the soft class target — tal.py:110-116
    align_metric = align_metric * mask_pos
    pos_align = align_metric.amax(dim=-1, keepdim=True)
    pos_overlaps = (overlaps * mask_pos).amax(dim=-1, keepdim=True)
    norm = (align_metric * pos_overlaps / (pos_align + eps)).amax(-2).unsqueeze(-1)
    return target_labels, target_bboxes, target_scores * norm, fg.bool(), target_gt_idx


def preprocess_targets(batch_idx: Tensor, cls: Tensor, bboxes: Tensor, batch_size: int, scale_xyxy: Tensor) -> Tensor:
    """(N,), (N,), (N,4 normalised xywh) -> (B, n_max, 5) [cls, x1, y1, x2, y2] in pixels, zero padded — loss.py:180-195."""
    if batch_idx.numel() == 0:
        return torch.zeros(batch_size, 0, 5)
    counts = [int((batch_idx == j).sum()) for j in range(batch_size)]
    out = torch.zeros(batch_size, max(counts), 5)
    for j in range(batch_size):
        sel = batch_idx == j
        if counts[j]:
            out[j, : counts[j], 0] = cls[sel].float()
            out[j, : counts[j], 1:] = bboxes[sel].float()
    out[..., 1:5] = xywh2xyxy(out[..., 1:5] * scale_xyxy)
    return out


def dfl_loss(pred_dist: Tensor, target: Tensor, reg_max: int = 16) -> Tensor:
    """DFLoss.__call__ — loss.py:73-88.  pred_dist (n*4, reg_max) logits, target (n, 4) distances in bins."""
    target = target.clamp(0, reg_max - 1 - 0.01)
    tl = target.long()
    tr = tl + 1
    wl = tr - target
    wr = 1 - wl
    ce_l = F.cross_entropy(pred_dist, tl.view(-1), reduction="none").view(tl.shape)
    ce_r = F.cross_entropy(pred_dist, tr.view(-1), reduction="none").view(tl.shape)
    return (ce_l * wl + ce_r * wr).mean(-1, keepdim=True)


def v8_detection_loss(feats: Sequence[Tensor], batch: Dict[str, Tensor], strides: Sequence[float], nc: int, reg_max: int = 16,
                      box_gain: float = 7.5, cls_gain: float = 0.5, dfl_gain: float = 1.5, tal_topk: int = 10, return_assign=False):
    """v8DetectionLoss.__call__ — loss.py:206-260.  feats: list of (B, 4*reg_max+nc, H_i, W_i) raw head outputs;
    batch: batch_idx (N,), cls (N,) or (N,1), bboxes (N,4) normalised xywh.  Returns (loss.sum()*B, loss_items[3])."""
    no = nc + 4 * reg_max
    B = feats[0].shape[0]
    cat = torch.cat([f.reshape(B, no, -1) for f in feats], 2)
    pred_distri, pred_scores = cat.split((4 * reg_max, nc), 1)
    pred_scores = pred_scores.permute(0, 2, 1).contiguous()  # (B,A,nc)
    pred_distri = pred_distri.permute(0, 2, 1).contiguous()  # (B,A,64)
    imgsz = torch.tensor(feats[0].shape[2:], dtype=torch.float32) * float(strides[0])  # (h, w)
    anchor_points, stride_tensor = make_anchors([f.shape[2:] for f in feats], strides, 0.5)
    targets = preprocess_targets(batch["batch_idx"].view(-1), batch["cls"].view(-1), batch["bboxes"], B, imgsz[[1, 0, 1, 0]])
    gt_labels, gt_bboxes = targets.split((1, 4), 2)
    mask_gt = (gt_bboxes.sum(2, keepdim=True) > 0).float()  # loss.py:229
    # decode predicted boxes in grid units — loss.py:197-204
    proj = torch.arange(reg_max, dtype=torch.float32)
    A = pred_distri.shape[1]
    dist = pred_distri.view(B, A, 4, reg_max).softmax(3).matmul(proj)
    pred_bboxes = dist2bbox(dist, anchor_points, xywh=False)
    tl, target_bboxes, target_scores, fg_mask, tgi = task_aligned_assign(
        pred_scores.detach().sigmoid(), (pred_bboxes.detach() * stride_tensor), anchor_points * stride_tensor, gt_labels,
        gt_bboxes, mask_gt, topk=tal_topk, num_classes=nc)
    tss = max(float(target_scores.sum()), 1.0)  # loss.py:247
    loss = torch.zeros(3)
    loss[1] = F.binary_cross_entropy_with_logits(pred_scores, target_scores.to(pred_scores.dtype), reduction="none").sum() / tss
    if fg_mask.sum():
        target_bboxes = target_bboxes / stride_tensor
        weight = target_scores.sum(-1)[fg_mask].unsqueeze(-1)  # loss.py:101
        iou = bbox_ciou(pred_bboxes[fg_mask], target_bboxes[fg_mask])
        loss[0] = ((1.0 - iou) * weight).sum() / tss
        ltrb = bbox2dist(anchor_points, target_bboxes, reg_max - 1)
        l_dfl = dfl_loss(pred_distri[fg_mask].view(-1, reg_max), ltrb[fg_mask], reg_max) * weight
        loss[2] = l_dfl.sum() / tss
    loss[0] *= box_gain
    loss[1] *= cls_gain
    loss[2] *= dfl_gain
    if return_assign:
        return loss.sum() * B, loss.detach(), dict(target_bboxes=target_bboxes, target_scores=target_scores, fg_mask=fg_mask,
                                                  target_gt_idx=tgi, pred_bboxes=pred_bboxes)
    return loss.sum() * B, loss.detach()


def synthetic_labels(batch: int, seed: int, n_mean: float = 12.0, nc: int = 10) -> Dict[str, Tensor]:
    """VisDrone-like synthetic labels: per image Poisson(n_mean) boxes clipped to [1, 300], uniform
    class, centres U(0.05, 0.95), sizes log-normal(median 0.03, sigma 0.6) clipped to [0.004, 0.5]; batch_idx sorted."""
    g = torch.Generator().manual_seed(seed)
    idx, cls, box = [], [], []
    for b in range(batch):
        n = int(torch.poisson(torch.tensor([n_mean]), generator=g).clamp(1, 300).item())
        idx.append(torch.full((n,), float(b)))
        cls.append(torch.randint(0, nc, (n,), generator=g).float())
        cxy = torch.rand(n, 2, generator=g) * 0.9 + 0.05
        wh = (torch.randn(n, 2, generator=g) * 0.6 + math.log(0.03)).exp().clamp(0.004, 0.5)
        box.append(torch.cat((cxy, wh), 1))
    return {"batch_idx": torch.cat(idx), "cls": torch.cat(cls).view(-1, 1), "bboxes": torch.cat(box)}
