"""CPU oracle of the predictor's input side for image sources — TEST INFRASTRUCTURE ONLY.

Restates (numpy): LetterBox.__call__ (ultralytics/data/augment.py:1545-1608), BasePredictor.pre_transform / preprocess
for non-tensor sources (engine/predictor.py:118-163): letterbox -> BGR->RGB -> HWC->CHW -> float / 255.

PARITY UNPINNED for the resize: LetterBox calls cv2.resize(img, new_unpad, interpolation=cv2.INTER_LINEAR)
(augment.py:1587) and cv2.copyMakeBorder; opencv-python (requirements.txt:11, `opencv-python>=4.6.0`, unpinned) is absent
from this container and the reference ships no fixture for it.  `resize_linear_u8` restates OpenCV's published 8-bit
bilinear algorithm (imgproc/resize.cpp: HResizeLinear with 11-bit fixed-point coefficients, VResizeLinear
`(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`, source coordinate (dx + 0.5) * scale - 0.5 clamped at the borders);
the no-resize case (frame already at the target size) involves no third-party arithmetic and is exact by construction.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _axis_coeffs(dst: int, src: int):
    """Per destination index: left source index and the two 11-bit coefficients (OpenCV resize.cpp, linear, 8U)."""
    scale = src / dst
    d = np.arange(dst, dtype=np.float64)
    f = (d + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    f = f - s
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    f = f.astype(np.float32)  # OpenCV keeps fx in float
    a0 = np.rint((1.0 - f) * COEF_SCALE).astype(np.int32)  # saturate_cast<short>(v * 2048): round half to even
    a1 = np.rint(f * COEF_SCALE).astype(np.int32)
    s1 = np.minimum(s + 1, src - 1)
    return s.astype(np.int64), s1.astype(np.int64), a0, a1


def resize_linear_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for HxWxC uint8."""
    h, w = img.shape[:2]
    x0, x1, ax0, ax1 = _axis_coeffs(new_w, w)
    y0, y1, by0, by1 = _axis_coeffs(new_h, h)
    src = img.astype(np.int32)
    hr = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]  # (h, new_w, c) int32, scaled by 2^11
    s0, s1 = hr[y0], hr[y1]
    out = (((by0[:, None, None] * (s0 >> 4)) >> 16) + ((by1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(shape: Tuple[int, int], new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True, stride=32):
    """(new_unpad_w, new_unpad_h, top, bottom, left, right) of LetterBox.__call__ — augment.py:1566-1591."""
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    elif scale_fill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
    if center:
        dw /= 2
        dh /= 2
    top, bottom = (int(round(dh - 0.1)) if center else 0), int(round(dh + 0.1))
    left, right = (int(round(dw - 0.1)) if center else 0), int(round(dw + 0.1))
    return new_unpad[0], new_unpad[1], top, bottom, left, right


def letterbox(img: np.ndarray, new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True, stride=32) -> np.ndarray:
    nw, nh, top, bottom, left, right = letterbox_geometry(img.shape[:2], new_shape, auto, scale_fill, scaleup, center, stride)
    if (img.shape[1], img.shape[0]) != (nw, nh):
        img = resize_linear_u8(img, nw, nh)
    out = np.full((nh + top + bottom, nw + left + right, img.shape[2]), 114, dtype=np.uint8)  # copyMakeBorder, value 114
    out[top : top + nh, left : left + nw] = img
    return out


def preprocess(frames, imgsz=(640, 640), auto=True, stride=32) -> np.ndarray:
    """BasePredictor.preprocess for a list of HWC BGR uint8 frames -> (N, 3, H, W) float32 in [0, 1] (predictor.py:125-135)."""
    same = len({f.shape for f in frames}) == 1
    im = np.stack([letterbox(f, imgsz, auto=auto and same, stride=stride) for f in frames])
    im = np.ascontiguousarray(im[..., ::-1].transpose((0, 3, 1, 2)))
    return im.astype(np.float32) / 255
