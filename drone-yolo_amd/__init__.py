"""drone-yolo_amd: the Drone-YOLO detection hot path on MI355X (gfx950).

Python host code on PyTorch-ROCm tensors calling hand-written HIP through the libdyolo C-ABI
(include/dyolo.h).  Public surface mirrors the reference (``from ultralytics import YOLO``):

    from drone_yolo_amd import YOLO
    model = YOLO("yolov8s-p2-repvgg.yaml")
    results = model.predict(torch.rand(8, 3, 640, 640), device=0)
"""
__version__ = "0.1.0"

from .engine.model import YOLO, Model  # noqa: E402
from .nn.tasks import DetectionModel  # noqa: E402

__all__ = ("YOLO", "Model", "DetectionModel", "__version__")
