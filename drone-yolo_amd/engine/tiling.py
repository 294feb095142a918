"""Tiled inference on frames larger than the model input.

The reference has no slicer of its own: ``mix6.py:84-89`` wraps the model in ``supervision.InferenceSlicer`` and
``examples/YOLOv8-SAHI-Inference-Video/yolov8_sahi.py:50-55`` uses ``sahi`` — neither package is vendored or installed, so
parity for this stage is unpinned and this module defines the behaviour: tiles of ``tile`` x ``tile`` pixels placed with a
stride of ``tile * (1 - overlap)``, the last tile of a row / column clamped to the frame edge; every tile goes through the
ordinary device pass (layout, model, decode, NMS) as one batch; kept boxes are shifted to frame coordinates and merged by
one more class-aware NMS (``merge_iou``).  All pixel and box work is on the device (``dy_tiles_u8_to_nchw_f32``,
``dy_rows_to_pred``, ``dy_nms``); the host only computes the tile offsets.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch

from .. import hip_ops as H
from .._lib import lib
from .predictor import DetectionPredictor
from .results import Results


def tile_offsets(h: int, w: int, tile: int, overlap: float = 0.2) -> List[Tuple[int, int]]:
    """(y, x) of every tile: stride tile*(1-overlap), the last one clamped so it ends at the edge (never starts below 0)."""
    step = max(int(round(tile * (1.0 - overlap))), 1)

    def axis(n):
        if n <= tile:
            return [0]
        out = list(range(0, n - tile, step)) + [n - tile]
        return sorted(set(out))

    return [(y, x) for y in axis(h) for x in axis(w)]


class TiledPredictor:
    def __init__(self, model, tile: int = 1280, overlap: float = 0.2, merge_iou: float = 0.7, max_det: int = 300, merge_max_det: int = 1000, **overrides):
        self.tile, self.overlap, self.merge_iou, self.merge_max_det = int(tile), float(overlap), float(merge_iou), int(merge_max_det)
        self.pred = DetectionPredictor(model, dict(max_det=max_det, **overrides))
        self.device = self.pred.device
        self.nc = model.yaml["nc"]

    def __call__(self, frame) -> Results:
        """frame: HWC BGR uint8 (numpy array or tensor).  Returns one Results in frame coordinates."""
        if not isinstance(frame, torch.Tensor):
            frame = torch.from_numpy(frame)
        if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
            raise ValueError("TiledPredictor expects one HWC uint8 frame")
        frame = frame.to(self.device).contiguous()
        hf, wf, _ = frame.shape
        offs = tile_offsets(hf, wf, self.tile, self.overlap)
        k = len(offs)
        offs_d = torch.tensor(offs, dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream().cuda_stream
        tiles = torch.empty((k, 3, self.tile, self.tile), dtype=torch.float32, device=self.device)
        H.check(lib().dy_tiles_u8_to_nchw_f32(frame.data_ptr(), offs_d.data_ptr(), tiles.data_ptr(), k, hf, wf, self.tile, self.tile, 1, 114.0, stream))
        self.pred.letterbox_info = None
        self.last_tiles = tiles  # the tile batch of the last frame (tests re-run the per-tile pass on it)
        cf = self.pred.forward_device(tiles)
        md = cf.nms.out.shape[1]
        merged_in = torch.empty((1, 4 + self.nc, k * md), dtype=torch.float32, device=self.device)
        H.check(lib().dy_rows_to_pred(cf.nms.out.data_ptr(), cf.nms.count.data_ptr(), offs_d.data_ptr(), merged_in.data_ptr(), k, md, self.nc, stream))
        a = self.pred.args
        merged = H.nms(merged_in, 0.0, self.merge_iou, max_det=self.merge_max_det, agnostic=bool(a["agnostic_nms"]), nc=self.nc)
        n = int(merged.count[0])
        boxes = merged.out[0, :n].clone()
        boxes[:, 0].clamp_(0, wf), boxes[:, 2].clamp_(0, wf), boxes[:, 1].clamp_(0, hf), boxes[:, 3].clamp_(0, hf)
        return Results(frame.cpu().numpy(), "frame.jpg", self.pred.model.names, boxes=boxes, orig_shape=(hf, wf))
