"""Result containers (reference: ultralytics/engine/results.py: ``Results``, ``Boxes`` :1004).

Only the tensor-view surface the detection path fills is kept (boxes); plotting / saving /
masks / keypoints are out of scope.  ``Boxes`` wraps the (n, 6) rows [x1, y1, x2, y2, conf, cls]
that ``dy_nms`` + ``dy_scale_boxes`` produced on the device.
"""
from __future__ import annotations

import numpy as np
import torch

from ..utils import ops


class BaseTensor:
    def __init__(self, data, orig_shape):
        self.data = data
        self.orig_shape = orig_shape

    @property
    def shape(self):
        return self.data.shape

    def cpu(self):
        return self if isinstance(self.data, np.ndarray) else self.__class__(self.data.cpu(), self.orig_shape)

    def numpy(self):
        return self if isinstance(self.data, np.ndarray) else self.__class__(self.data.cpu().numpy(), self.orig_shape)

    def cuda(self):
        return self.__class__(torch.as_tensor(self.data).cuda(), self.orig_shape)

    def to(self, *args, **kwargs):
        return self.__class__(torch.as_tensor(self.data).to(*args, **kwargs), self.orig_shape)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.__class__(self.data[idx], self.orig_shape)


class Boxes(BaseTensor):
    """Detections of one image — reference results.py:1004-1170."""

    def __init__(self, boxes, orig_shape):
        if boxes.ndim == 1:
            boxes = boxes[None, :]
        assert boxes.shape[-1] in {6, 7}, f"expected 6 or 7 values but got {boxes.shape[-1]}"
        super().__init__(boxes, orig_shape)
        self.is_track = boxes.shape[-1] == 7

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, -2]

    @property
    def cls(self):
        return self.data[:, -1]

    @property
    def id(self):
        return self.data[:, -3] if self.is_track else None

    @property
    def xywh(self):
        return ops.xyxy2xywh(self.xyxy)

    @property
    def xyxyn(self):
        xyxy = self.xyxy.clone() if isinstance(self.xyxy, torch.Tensor) else np.copy(self.xyxy)
        xyxy[..., [0, 2]] /= self.orig_shape[1]
        xyxy[..., [1, 3]] /= self.orig_shape[0]
        return xyxy

    @property
    def xywhn(self):
        xywh = ops.xyxy2xywh(self.xyxy)
        xywh[..., [0, 2]] /= self.orig_shape[1]
        xywh[..., [1, 3]] /= self.orig_shape[0]
        return xywh


class Results:
    """Per-image result — reference results.py (`Results`): orig_img, orig_shape, boxes, names, path, speed."""

    def __init__(self, orig_img, path, names, boxes=None, speed=None, orig_shape=None):
        self._orig_img = orig_img
        self.orig_shape = tuple(orig_shape) if orig_shape is not None else tuple(orig_img.shape[:2])
        self.boxes = Boxes(boxes, self.orig_shape) if boxes is not None else None
        self.masks = self.probs = self.keypoints = self.obb = None
        self.speed = speed if speed is not None else {"preprocess": None, "inference": None, "postprocess": None}
        self.names = names
        self.path = path

    @property
    def orig_img(self):
        """HWC uint8 image; for tensor sources converted lazily (reference ops.py:841-851 does it eagerly)."""
        im = self._orig_img
        if isinstance(im, torch.Tensor):
            im = (im.permute(1, 2, 0).contiguous() * 255).clamp(0, 255).to(torch.uint8).cpu().numpy()
            self._orig_img = im
        return im

    def __len__(self):
        return 0 if self.boxes is None else len(self.boxes)

    def cpu(self):
        r = Results(self._orig_img, self.path, self.names, None, self.speed, self.orig_shape)
        r.boxes = self.boxes.cpu() if self.boxes is not None else None
        return r

    def numpy(self):
        r = Results(self._orig_img, self.path, self.names, None, self.speed, self.orig_shape)
        r.boxes = self.boxes.numpy() if self.boxes is not None else None
        return r

    def to(self, *args, **kwargs):
        r = Results(self._orig_img, self.path, self.names, None, self.speed, self.orig_shape)
        r.boxes = self.boxes.to(*args, **kwargs) if self.boxes is not None else None
        return r

    def __getitem__(self, idx):
        r = Results(self._orig_img, self.path, self.names, None, self.speed, self.orig_shape)
        r.boxes = self.boxes[idx] if self.boxes is not None else None
        return r

    # ---- text forms of a detection result (reference results.py:633-666 verbose, :668-757 save_txt, :759-823 summary, :906-940 to_json) ----
    def verbose(self) -> str:
        """'2 persons, 1 car, ' — detections per class in ascending class order; '(no detections), ' for none."""
        if len(self) == 0:
            return "(no detections), "
        cls = torch.as_tensor(self.boxes.cls).cpu()
        out = ""
        for c in cls.unique():
            n = int((cls == c).sum())
            out += f"{n} {self.names[int(c)]}{'s' * (n > 1)}, "
        return out

    def summary(self, normalize: bool = False, decimals: int = 5) -> list:
        """One dict per detection: name, class, confidence, box {x1, y1, x2, y2} (divided by the image size with ``normalize``)."""
        if self.boxes is None:
            return []
        h, w = self.orig_shape if normalize else (1, 1)
        rows = torch.as_tensor(self.boxes.data).cpu().tolist()
        out = []
        for r in rows:
            k = int(r[-1])
            d = {"name": self.names[k], "class": k, "confidence": round(r[-2], decimals),
                 "box": {"x1": round(r[0] / w, decimals), "y1": round(r[1] / h, decimals), "x2": round(r[2] / w, decimals), "y2": round(r[3] / h, decimals)}}
            if self.boxes.is_track:
                d["track_id"] = int(r[-3])
            out.append(d)
        return out

    def to_json(self, normalize: bool = False, decimals: int = 5) -> str:
        import json

        return json.dumps(self.summary(normalize=normalize, decimals=decimals), indent=2)

    tojson = to_json

    def save_txt(self, txt_file, save_conf: bool = False) -> str:
        """Append one line per detection, `class x_center y_center width height [confidence]` with the box normalised to the image (the label format)."""
        from pathlib import Path

        lines = []
        if self.boxes is not None and len(self.boxes):
            xywhn = torch.as_tensor(self.boxes.xywhn).cpu().tolist()
            conf = torch.as_tensor(self.boxes.conf).cpu().tolist()
            cls = torch.as_tensor(self.boxes.cls).cpu().tolist()
            for b, s, c in zip(xywhn, conf, cls):
                vals = (int(c), *b) + ((s,) if save_conf else ())
                lines.append(("%g " * len(vals)).rstrip() % vals)
        if lines:
            Path(txt_file).parent.mkdir(parents=True, exist_ok=True)
            with open(txt_file, "a") as f:
                f.writelines(t + "\n" for t in lines)
        return str(txt_file)
