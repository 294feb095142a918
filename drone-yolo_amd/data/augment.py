"""LetterBox of the predictor's input side (reference: ultralytics/data/augment.py:1486-1620), device version.

Same constructor arguments and the same host geometry as the reference (augment.py:1566-1591); the pixels are produced by
``dy_letterbox_u8_to_nchw_f32``, which also does what BasePredictor.preprocess does next for non-tensor sources
(BGR->RGB, HWC->CHW, float, /255 — engine/predictor.py:125-135), so the output is the model's fp32 NCHW input.
Label updating (``_update_labels``) belongs to the training data pipeline and is not part of the accelerated path.
"""
from __future__ import annotations

from typing import Tuple

import torch

from .. import hip_ops as H


class LetterBox:
    def __init__(self, new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True, stride=32):
        self.new_shape = (new_shape, new_shape) if isinstance(new_shape, int) else tuple(new_shape)
        self.auto, self.scale_fill, self.scaleup, self.center, self.stride = auto, scale_fill, scaleup, center, int(stride)

    def geometry(self, shape: Tuple[int, int]):
        """(new_w, new_h, top, bottom, left, right) for a (h, w) frame — augment.py:1566-1591."""
        new_shape = self.new_shape
        r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
        if not self.scaleup:
            r = min(r, 1.0)
        new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
        dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
        if self.auto:
            dw, dh = dw % self.stride, dh % self.stride
        elif self.scale_fill:
            dw, dh = 0.0, 0.0
            new_unpad = (new_shape[1], new_shape[0])
        if self.center:
            dw /= 2
            dh /= 2
        top, bottom = (int(round(dh - 0.1)) if self.center else 0), int(round(dh + 0.1))
        left, right = (int(round(dw - 0.1)) if self.center else 0), int(round(dw + 0.1))
        return new_unpad[0], new_unpad[1], top, bottom, left, right

    def into(self, frame: torch.Tensor, out: torch.Tensor, swap_rb: bool = True) -> None:
        """ONE device uint8 (H, W, 3) frame -> ``out``, a (1, 3, Hn, Wn) fp32 slice of a batch tensor whose size is this LetterBox's output
        for the frame: the per-image form ``pre_transform`` uses for sources of different shapes (engine/predictor.py:147-163, auto = False:
        every image letterboxed to the full ``new_shape``)."""
        h0, w0, _ = frame.shape
        nw, nh, top, bottom, left, right = self.geometry((h0, w0))
        if tuple(out.shape) != (1, 3, nh + top + bottom, nw + left + right) or not out.is_contiguous():
            raise ValueError(f"LetterBox.into: out must be a contiguous (1, 3, {nh + top + bottom}, {nw + left + right}) slice")
        H.letterbox(frame[None], nw, nh, top, left, nh + top + bottom, nw + left + right, swap_rb, out=out)

    def __call__(self, frames: torch.Tensor, swap_rb: bool = True) -> torch.Tensor:
        """frames: device uint8 (N, H, W, 3) -> fp32 (N, 3, Hn, Wn) in [0, 1], letterboxed."""
        H.require_device(frames, "frames")
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3 or not frames.is_contiguous():
            raise ValueError("LetterBox expects a contiguous uint8 (N, H, W, 3) device tensor")
        n, h0, w0, _ = frames.shape
        nw, nh, top, bottom, left, right = self.geometry((h0, w0))
        return H.letterbox(frames, nw, nh, top, left, nh + top + bottom, nw + left + right, swap_rb)
