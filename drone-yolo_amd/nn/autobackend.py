"""``AutoBackend`` of the accelerated path (reference: ultralytics/nn/autobackend.py:80-772).

The reference's AutoBackend picks one of seventeen runtimes; the Drone-YOLO path has one — the in-memory module graph on
libdyolo kernels — so this class keeps the reference's constructor arguments, attributes and duties for the ``nn_module`` /
``*.pt`` cases (autobackend.py:143-170): move to the device, ``fuse()``, choose the storage precision (``fp16`` -> half, the
package's ``dtype`` extension -> bf16 / fp16 / fp32), freeze the parameters, expose ``stride`` / ``names`` / ``fp16``;
``forward`` (autobackend.py:535-556) runs the executor, ``warmup`` (autobackend.py:759-772) runs one dummy pass so the
launch plan of that shape is recorded (and the weight packs built) before the first real batch.
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Union

import torch
import torch.nn as nn


class AutoBackend(nn.Module):
    def __init__(self, weights: Union[str, Path, nn.Module] = "yolov8s-p2-repvgg.yaml", device: Optional[torch.device] = None, dnn: bool = False,
                 data=None, fp16: bool = False, batch: int = 1, fuse: bool = True, verbose: bool = True, dtype: Optional[torch.dtype] = None):
        super().__init__()
        from ..utils.torch_utils import select_device

        if dnn:
            raise NotImplementedError("dnn=True (OpenCV DNN / ONNX) is an export runtime: outside the accelerated path")
        device = select_device("" if device is None else device)
        if isinstance(weights, nn.Module):
            model = weights
        else:
            w = str(weights)
            if w.endswith(".pt"):
                from .checkpoint import load_reference_checkpoint

                model, _ = load_reference_checkpoint(w)
            elif w.endswith((".yaml", ".yml")):
                from .tasks import DetectionModel

                model = DetectionModel(w, verbose=False)
            else:
                raise NotImplementedError(f"'{w}': exported formats (onnx, engine, ...) are outside the accelerated path; give a module, *.pt or *.yaml")
        model = model.to(device)
        if fuse:
            model = model.fuse(verbose=verbose)  # BatchNorm folding happens when the packs are built: this drops stale ones
        self.model = model.eval()
        self.model.requires_grad_(False)
        self.device, self.fp16 = device, bool(fp16) or dtype == torch.float16
        if dtype is None:  # the predictor's rule: half -> float16, otherwise the bar-exact precision (engine/predictor.py::EXACT_DTYPE)
            from ..engine.predictor import resolve_dtype

            dtype = resolve_dtype(None, bool(fp16), model)
        self.dtype = dtype
        self.stride = max(int(model.stride.max()), 32)
        self.names = model.names
        self.pt = self.nn_module = True
        self.jit = self.onnx = self.engine = self.triton = False
        self.batch, self.task, self.end2end = batch, "detect", getattr(model, "end2end", False)

    def forward(self, im: torch.Tensor, augment: bool = False, visualize: bool = False, embed=None):
        """fp32 NCHW image batch on the device -> (decoded (N, 4 + nc, A) fp32, raw maps or None)."""
        if augment or visualize or embed:
            raise NotImplementedError("augment / visualize / embed are outside the accelerated path")
        return self.model._predict_once(im.to(self.device, torch.float32).contiguous(), image_dtype=self.dtype)

    def warmup(self, imgsz=(1, 3, 640, 640)) -> None:
        im = torch.zeros(*imgsz, dtype=torch.float32, device=self.device)
        self.forward(im)
        torch.cuda.synchronize(self.device)
