"""``YOLO`` / ``Model`` user API (reference: ultralytics/engine/model.py:29-1177,
ultralytics/models/yolo/model.py:11-59) for the detection task on the MI355X path."""
from __future__ import annotations

from pathlib import Path
from typing import List, Union

import torch
import torch.nn as nn

from ..nn.tasks import DetectionModel, yaml_model_load
from ..utils import LOGGER
from .predictor import DetectionPredictor
from .results import Results


class Model(nn.Module):
    """Same constructor and call surface as the reference ``Model`` (engine/model.py:82-87, 501-560, 744-817)."""

    def __init__(self, model: Union[str, Path] = "yolov8s-p2-repvgg.yaml", task: str = None, verbose: bool = False) -> None:
        super().__init__()
        self.predictor = None
        self.model = None
        self.trainer = None
        self.ckpt = None
        self.cfg = None
        self.ckpt_path = None
        self.overrides = {}
        self.task = task or "detect"
        if self.task != "detect":
            raise NotImplementedError(f"task '{self.task}': only 'detect' is on the accelerated path")
        model = str(model).strip()
        if Path(model).suffix in {".yaml", ".yml"}:
            self._new(model, verbose=verbose)
        elif Path(model).suffix == ".pt":
            self._load(model)
        else:
            raise NotImplementedError(f"'{model}': give a model YAML (*.yaml) or a state-dict checkpoint (*.pt); "
                                      "weight-name downloads need network access and are out of scope")
        self.model_name = model

    def _new(self, cfg: str, task=None, model=None, verbose=False) -> None:
        """Build from a YAML — reference engine/model.py:231-264."""
        cfg_dict = yaml_model_load(cfg)
        self.cfg = cfg
        self.model = (model or DetectionModel)(cfg_dict, verbose=verbose)
        self.overrides["model"] = self.cfg
        self.overrides["task"] = self.task

    def _load(self, weights: str, task=None) -> None:
        """Load a checkpoint: either {'yaml': cfg dict, 'model': state_dict} as written by ``save``, or the reference's
        pickled module graph (``ema`` / ``model`` entries, tasks.py:786-926) through the restricted unpickler of
        nn/checkpoint.py — no ``ultralytics`` package needed."""
        from ..nn.checkpoint import load_reference_checkpoint

        self.model, meta = load_reference_checkpoint(str(weights))
        self.ckpt, self.ckpt_path = meta, weights
        self.overrides["model"] = weights

    def save(self, filename: Union[str, Path] = "saved_model.pt") -> None:
        """Reference engine/model.py:366-395: a checkpoint with the trainer's key layout ({epoch, ema / model, optimizer,
        train_args, date, version ...}); the module graph is pickled in fp16 under the reference's class paths
        (nn/checkpoint.py::save_reference_checkpoint), so both this package and the reference itself can load it."""
        from datetime import datetime

        from ..nn.checkpoint import save_reference_checkpoint

        extra = {k: v for k, v in (self.ckpt or {}).items() if k in ("epoch", "best_fitness", "updates", "train_args", "train_metrics", "train_results")}
        extra["date"] = datetime.now().isoformat()
        save_reference_checkpoint(filename, self.model, self.model.state_dict(), extra=extra)

    def __call__(self, source=None, stream: bool = False, **kwargs):
        return self.predict(source, stream, **kwargs)

    def predict(self, source=None, stream: bool = False, predictor=None, **kwargs) -> List[Results]:
        """Reference engine/model.py:501-560: predictor created on first use, conf defaults to 0.25."""
        if source is None:
            raise ValueError("'source' is missing; the device path takes a BCHW float tensor in [0, 1]")
        args = {**self.overrides, "conf": 0.25, **kwargs}
        args.pop("model", None), args.pop("task", None), args.pop("mode", None)
        if self.predictor is None or getattr(self, "_pred_args", None) != args:
            self.predictor = (predictor or DetectionPredictor)(self.model, overrides=args)
            self._pred_args = args
        return self.predictor(source, stream=stream)

    def profile(self, source, **kwargs) -> list:
        """Per-layer device time of one pass over ``source`` (reference ``predict(profile=True)`` -> ``_profile_one_layer``, nn/tasks.py:171-191):
        [{layer, type, launches, ms, kernels}], see ``DetectionPredictor.profile_layers``."""
        self.predict(source, **kwargs)
        return self.predictor.profile_layers(self.predictor.preprocess(source))

    def train(self, trainer=None, **kwargs):
        """Reference engine/model.py:744-817: build the trainer from the model + overrides, train, then continue with the
        trained weights.  ``data``: a tensor dataset (.pt / dict) or "synthetic[:N]" (engine/trainer.py::load_dataset);
        ``device="0,1,.."`` trains with one rank per GPU (child processes under torch.distributed.run, RCCL all-reduce),
        after which rank-less this process reloads ``weights/last.pt`` — as the reference does (model.py:806-813)."""
        from .trainer import DetectionTrainer

        args = {**{k: v for k, v in self.overrides.items() if k not in ("task", "mode")}, **kwargs}
        devs = [x for x in str(args.get("device", "")).replace("cuda:", "").split(",") if x.strip() != ""]
        multi = len(devs) > 1
        if multi:
            # the ranks rebuild the model from a file: hand them the current weights
            import tempfile

            start = Path(tempfile.mkdtemp(prefix="dyolo_train_")) / "start.pt"
            self.save(start)
            args["model"] = str(start)
            self.trainer = (trainer or DetectionTrainer)(overrides=args)
        else:
            args.setdefault("model", self.overrides.get("model"))
            self.trainer = (trainer or DetectionTrainer)(self.model, overrides=args)
        out = self.trainer.train()
        last = self.trainer.last
        if multi and last.exists():
            self._load(str(last))
        elif not multi:
            # the reference continues with best/last.pt = the EMA weights whatever the GPU count (model.py:812-814); the live model holds
            # the raw optimizer weights (flat-buffer views), so the EMA state is loaded into it (same values last.pt carries, fp32)
            self.model.load_state_dict(self.trainer.ema.state_dict(self.model))
            self.model.eval()  # packs are dropped by train() -> eval()
        self.predictor = None
        return out

    def val(self, validator=None, **kwargs) -> dict:
        """Reference engine/model.py:620-656 (``Model.val``): the model's own weights scored on a dataset by the task's validator
        (models/yolo/detect/val.py; engine/validator.py here: the model pass and the ``multi_label`` NMS at conf 0.001 on the device, matching and
        AP on the host).  ``data``: a tensor dataset — a dict / ``.pt`` in the training layout, its ``"val"`` split when it has one — or
        ``"synthetic[:N]"`` (engine/trainer.py::load_dataset; image folders and dataset YAMLs are outside the accelerated path);
        ``batch``, ``imgsz``, ``conf``, ``iou``, ``max_det``, ``half`` / ``dtype``, ``device`` as the reference's arguments.  Returns the
        reference's ``results_dict`` (metrics/precision(B) ... metrics/mAP50-95(B), fitness) with the validation losses; kept in ``self.metrics``."""
        from .predictor import resolve_dtype
        from .trainer import TensorLoader, load_dataset
        from .validator import DetectionValidator
        from ..utils.torch_utils import select_device

        args = {**{k: v for k, v in self.overrides.items() if k not in ("task", "mode")}, **kwargs}
        if args.get("data") is None:
            raise ValueError("val(): 'data' is missing (a tensor dataset: dict / .pt, or 'synthetic[:N]')")
        device = select_device(args.get("device", ""))
        data = load_dataset(args["data"], int(args.get("imgsz", 640)), self.model.yaml["nc"], int(args.get("seed", 0)))
        data = data.get("val") or data
        dtype = resolve_dtype(args.get("dtype"), bool(args.get("half", False)), self.model)
        if dtype not in (torch.float32, torch.float16, torch.bfloat16):
            dtype = torch.float32  # (the validator's pass returns the raw maps for the loss: the storage types of the training path)
        was_training = self.model.training
        model = self.model.to(device)
        loader = TensorLoader({k: data[k] for k in ("img", "batch_idx", "cls", "bboxes")}, int(args.get("batch") or 16), 0, 1, shuffle=False)
        try:
            self.metrics = (validator or DetectionValidator)(args)(model, loader, device, dtype)
        finally:
            model.train(was_training)
        self.predictor = None  # (the pass may have re-packed weights for another storage type)
        return self.metrics

    def fuse(self):
        self.model.fuse()
        return self

    def info(self, detailed: bool = False, verbose: bool = True):
        return self.model.info(detailed=detailed, verbose=verbose)

    @property
    def names(self):
        return self.model.names

    @property
    def device(self):
        return next(self.model.parameters()).device

    def to(self, device):
        self.model.to(device)
        return self


class YOLO(Model):
    """``YOLO(model, task=None, verbose=False)`` — reference models/yolo/model.py:11-23."""

    def __init__(self, model="yolov8s-p2-repvgg.yaml", task=None, verbose=False):
        super().__init__(model=model, task=task, verbose=verbose)

    @property
    def task_map(self):
        return {"detect": {"model": DetectionModel, "predictor": DetectionPredictor}}
