"""Post-processing ops of the detection path (reference: ultralytics/utils/ops.py).

``non_max_suppression`` (:181-332) keeps the reference signature and return type; the work is
done by ``dy_nms`` (filter -> sort -> greedy suppression on the device).  ``xywh2xyxy`` (:432-449),
``scale_boxes`` (:92-127), ``clip_boxes`` (:335-354), ``make_divisible`` (:130-143) and ``Profile``
(:17-62) are the small host helpers around it.
"""
from __future__ import annotations

import contextlib
import math
import time
from typing import List, Optional

import numpy as np
import torch

from .. import hip_ops as H


class Profile(contextlib.ContextDecorator):
    """Wall-clock timer that synchronises the device around the block — reference ops.py:17-62."""

    def __init__(self, t=0.0, device: Optional[torch.device] = None):
        self.t = t
        self.device = device
        self.cuda = bool(device and str(device).startswith("cuda"))

    def __enter__(self):
        self.start = self.time()
        return self

    def __exit__(self, type, value, traceback):
        self.dt = self.time() - self.start
        self.t += self.dt

    def __str__(self):
        return f"Elapsed time is {self.t} s"

    def time(self):
        if self.cuda:
            torch.cuda.synchronize(self.device)
        return time.perf_counter()


def make_divisible(x, divisor):
    """Nearest multiple of divisor not below x — reference ops.py:130-143."""
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def empty_like(x):
    return torch.empty_like(x, dtype=torch.float32) if isinstance(x, torch.Tensor) else np.empty_like(x, dtype=np.float32)


def xywh2xyxy(x):
    """(cx, cy, w, h) -> (x1, y1, x2, y2); output is always fp32 — reference ops.py:432-449."""
    assert x.shape[-1] == 4, f"input shape last dimension expected 4 but input shape is {x.shape}"
    y = empty_like(x)
    xy = x[..., :2]
    wh = x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def xyxy2xywh(x):
    assert x.shape[-1] == 4, f"input shape last dimension expected 4 but input shape is {x.shape}"
    y = empty_like(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def clip_boxes(boxes, shape):
    """Clamp xyxy boxes to (h, w) — reference ops.py:335-354."""
    if isinstance(boxes, torch.Tensor):
        boxes[..., 0] = boxes[..., 0].clamp(0, shape[1])
        boxes[..., 1] = boxes[..., 1].clamp(0, shape[0])
        boxes[..., 2] = boxes[..., 2].clamp(0, shape[1])
        boxes[..., 3] = boxes[..., 3].clamp(0, shape[0])
    else:
        boxes[..., [0, 2]] = boxes[..., [0, 2]].clip(0, shape[1])
        boxes[..., [1, 3]] = boxes[..., [1, 3]].clip(0, shape[0])
    return boxes


def letterbox_params(img1_shape, img0_shape, ratio_pad=None):
    """(gain, pad_x, pad_y) that undo a letterbox from img0 to img1 — the scalar part of ops.py:109-117."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    return gain, pad[0], pad[1]


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True, xywh=False):
    """Host/tensor helper with the reference semantics (ops.py:92-127); the predictor uses the
    fused device version ``dy_scale_boxes`` on the NMS output instead."""
    gain, px, py = letterbox_params(img1_shape, img0_shape, ratio_pad)
    if padding:
        boxes[..., 0] -= px
        boxes[..., 1] -= py
        if not xywh:
            boxes[..., 2] -= px
            boxes[..., 3] -= py
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


def non_max_suppression(
    prediction,
    conf_thres=0.25,
    iou_thres=0.45,
    classes=None,
    agnostic=False,
    multi_label=False,
    labels=(),
    max_det=300,
    nc=0,
    max_time_img=0.05,
    max_nms=30000,
    max_wh=7680,
    in_place=True,
    rotated=False,
    end2end=False,
    return_padded=False,
) -> List[torch.Tensor]:
    """Reference ops.py:181-332 on the device via ``dy_nms``: the predictor's single-label path and the validator's ``multi_label``
    path (one candidate per (anchor, class) pair above conf, ops.py:286-288; models/yolo/detect/val.py:93-106).

    Returns a list (one (n_i, 6) tensor [x1, y1, x2, y2, conf, cls] per image) like the reference;
    ``return_padded=True`` returns the device-resident ``NmsBuffers`` (out (N,max_det,6), count (N,),
    index (N,max_det)) without any host synchronisation.  ``max_time_img`` is accepted and ignored:
    the wall-clock break (ops.py:328-330) is non-deterministic and the device kernel has no need for it.
    """
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if labels or rotated or end2end or prediction.shape[-1] == 6:
        raise NotImplementedError("labels (autolabelling) / rotated / end2end NMS variants are not on the accelerated path")
    H.require_device(prediction, "prediction")
    nc = nc or (prediction.shape[1] - 4)
    mask = None
    if classes is not None:
        mask = torch.zeros(nc, dtype=torch.uint8)
        mask[torch.as_tensor(list(classes), dtype=torch.long)] = 1
        mask = mask.to(prediction.device)
    pred = prediction if (prediction.dtype == torch.float32 and prediction.is_contiguous()) else prediction.float().contiguous()
    bufs = H.nms(pred, float(conf_thres), float(iou_thres), max_det=int(max_det), max_nms=int(max_nms),
                 max_wh=float(max_wh), agnostic=bool(agnostic), nc=int(nc), classes_mask=mask, multi_label=bool(multi_label))
    if return_padded:
        return bufs
    counts = bufs.count.tolist()  # one device->host sync for the whole batch
    return [bufs.out[i, :k].clone() for i, k in enumerate(counts)]
