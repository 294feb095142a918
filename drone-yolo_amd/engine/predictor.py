"""Inference loop of the detection path on the device.

Reference: ``BasePredictor`` (ultralytics/engine/predictor.py:118-323: preprocess, inference,
stream_inference, setup_model) and ``DetectionPredictor`` (models/yolo/detect/predict.py:23-73:
postprocess = NMS + scale_boxes + Results).  MI355X-first differences:

* one pass = input layout/dtype conversion -> 80-odd conv launches -> decode -> NMS -> box rescale,
  all libdyolo kernels on one HIP stream, recorded once per input shape as a ``LaunchPlan`` and
  replayed (optionally from a hipGraph) afterwards — no per-image Python loop, no host syncs
  inside the pass (the reference syncs at every boolean index of ops.py:266-330).
* results stay on the device as padded (N, max_det, 6) + counts; ``Results`` are materialised with
  one device->host transfer of the counts.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Tuple

import torch

from .. import hip_ops as H
from ..utils import LOGGER, ops
from ..utils.torch_utils import select_device
from ..nn.autobackend import AutoBackend
from .results import Results

_DTYPE_NAMES = {"f16x2": H.F16X2, "split": H.F16X2, "fp8": H.FP8, "float8_e4m3fn": H.FP8, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "half": torch.float16,
                "float16": torch.float16, "fp32": torch.float32, "float32": torch.float32}


# The precision a predictor runs in when the caller names none.  The reference's default is ``half: False`` = fp32 (cfg/default.yaml:54,
# engine/predictor.py:315-322), and the drop-in's default must reproduce ITS detections: class / index exact, IoU >= 0.999 (BASELINE.json).
# ``half=True`` selects float16 storage as in the reference; bf16 and fp8 only on request (they do not meet that bar).
# Two precisions meet it: fp32 storage on the fp32 MFMA (1/16 of the 16-bit rate) and split float16 (hip_ops.F16X2: every value a float16
# pair hi + lo 2^-11, three 16-bit MFMAs per product — kept sets identical to the reference's on every fixture, at ~2x the fp32 path's
# throughput).  The default is the split type wherever its kernels cover the model (dense convolutions on whole groups of 8 channels),
# fp32 otherwise (the DWConv of the -sf YAML).
EXACT_DTYPE = H.F16X2


def split_supported(model) -> bool:
    """Whether every convolution of ``model`` is one the split-float16 kernels take (dense, channel counts in whole groups of 8 — the image
    layer's input and the class logits excepted)."""
    import torch.nn as nn

    convs = [m for k, m in model.named_modules() if isinstance(m, nn.Conv2d) and ".dfl" not in k]  # (DFL's fixed 16 -> 1 conv lives inside the decode kernel)
    for c in convs:
        if c.groups != 1:  # DWConv of the -sf YAML: the small-group kernel (one output channel per group, 1 / 2 / 4 inputs each)
            if c.out_channels != c.groups or c.in_channels // c.groups not in (1, 2, 4) or c.out_channels % 8 or c.kernel_size[0] ** 2 * (c.in_channels // c.groups) * c.out_channels * 4 > 48 * 1024:
                return False
            continue
        if c.kernel_size not in ((1, 1), (3, 3)) or c.stride[0] not in (1, 2):
            return False
        if (c.in_channels % 8 and c.in_channels > 3) or (c.out_channels % 8 and c.bias is None):  # (plain biased 1x1 = Detect's fp32 outputs)
            return False
        if c.kernel_size == (3, 3) and c.in_channels > 680:  # the 3x3 tap table of csrc/conv_gemm_fk.hip (1,536 words)
            return False
    return bool(convs)


def resolve_dtype(dtype=None, half: bool = False, model=None) -> torch.dtype:
    if isinstance(dtype, torch.dtype):
        return dtype
    if dtype is None:
        if half:
            return torch.float16
        return EXACT_DTYPE if (model is None or split_supported(model)) else torch.float32
    try:
        return _DTYPE_NAMES[str(dtype).lower()]
    except KeyError:
        raise ValueError(f"unknown dtype '{dtype}', expected one of {sorted(_DTYPE_NAMES)}") from None


class CompiledForward:
    """A recorded forward for one (batch, H, W): its launch plan, static buffers and optional hipGraph."""

    def __init__(self):
        self.plan = H.LaunchPlan()
        self.pred: Optional[torch.Tensor] = None  # decoded (N, 4+nc, A) fp32
        self.nms: Optional[H.NmsBuffers] = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.graph_tail: Optional[torch.cuda.CUDAGraph] = None  # every launch but the first (the one that reads the image): see DetectionPredictor._capture
        self.static_in: Optional[torch.Tensor] = None
        self.weights_sig = None  # BaseModel.weights_signature() of the weights this plan baked in


class DetectionPredictor:
    """Device-resident predictor (reference predictor.py:64-323, detect/predict.py:8-73)."""

    max_compiled = 8  # recorded (batch, H, W) passes kept alive (each holds activations, NMS buffers and a graph: a video's ragged last batch must not pile up)

    def __init__(self, model, overrides: Optional[dict] = None):
        a = dict(conf=0.25, iou=0.7, max_det=300, classes=None, agnostic_nms=False, half=False, dtype=None, device="",
                 verbose=False, graph=True, max_nms=30000, max_wh=7680, imgsz=640, fp8_layers=None, batch=None, augment=False)
        a.update(overrides or {})
        self.args = a
        self.device = select_device(a["device"])
        # dtype="fp8-mixed" (BASELINE config 5): float16 storage with the internals of the C2f blocks the error budget allows in
        # e4m3 (BaseModel.fp8_plan_off_p2, or the blocks given as overrides["fp8_layers"]); dtype="fp8": the whole trunk in e4m3, Detect tail float16
        self.mixed8 = isinstance(a["dtype"], str) and a["dtype"].lower().replace("_", "-") == "fp8-mixed"
        self.dtype = torch.float16 if self.mixed8 else resolve_dtype(a["dtype"], a["half"], model)
        # setup_model (predictor.py:300-323): AutoBackend moves the graph to the device, fuses, picks the precision and freezes it
        self.backend = AutoBackend(model, device=self.device, fp16=bool(a["half"]), dtype=self.dtype, fuse=False, verbose=bool(a["verbose"]))
        self.model = self.backend.model
        self._compiled: Dict[Tuple, CompiledForward] = {}
        self._classes_mask = None
        if a["classes"] is not None:
            m = torch.zeros(self.model.yaml["nc"], dtype=torch.uint8)
            m[torch.as_tensor(list(a["classes"]), dtype=torch.long)] = 1
            self._classes_mask = m.to(self.device)

    # ---- stages (names follow the reference) -------------------------------------------------------
    def preprocess(self, im) -> torch.Tensor:
        """Tensor source: BCHW float in [0,1], moved to the device (predictor.py:118-136; tensors are not /255).
        Array source — a list of HWC BGR uint8 numpy frames of ONE shape, or a uint8 (N, H, W, 3) tensor: LetterBox
        (``auto`` = minimum rectangle, as pre_transform picks for same-shape batches, predictor.py:147-163), BGR->RGB,
        HWC->CHW and /255 in one kernel; ``self.letterbox_info`` then holds what postprocess needs to map boxes back."""
        self.letterbox_info = None
        if isinstance(im, (list, tuple)) or (isinstance(im, torch.Tensor) and im.dtype == torch.uint8 and im.dim() == 4 and im.shape[-1] == 3) \
                or type(im).__name__ == "ndarray":
            import numpy as np

            from ..data.augment import LetterBox

            if not isinstance(im, torch.Tensor):
                frames = [im] if type(im).__name__ == "ndarray" and im.ndim == 3 else list(im)
                if len({f.shape for f in frames}) != 1:
                    # sources of different shapes (pre_transform, predictor.py:147-163: same_shapes False -> auto False): every image is
                    # letterboxed on its own to the full imgsz x imgsz and the batch stacked; boxes map back per image (r05)
                    lb = LetterBox(self.args.get("imgsz", 640), auto=False, stride=int(self.model.stride.max()))
                    size = lb.new_shape
                    out = torch.empty((len(frames), 3, size[0], size[1]), dtype=torch.float32, device=self.device)
                    info = []
                    for i, f in enumerate(frames):
                        if f.ndim != 3 or f.shape[2] != 3 or f.dtype != np.uint8:
                            raise ValueError("image sources must be uint8 HWC BGR frames")
                        lb.into(torch.from_numpy(np.ascontiguousarray(f)).to(self.device), out[i : i + 1], swap_rb=True)
                        info.append((f.shape[0], f.shape[1], size[0], size[1]))
                    self.letterbox_info = info
                    return out
                im = torch.from_numpy(np.ascontiguousarray(np.stack(frames)))
            frames = im.to(self.device).contiguous()
            lb = LetterBox(self.args.get("imgsz", 640), auto=True, stride=int(self.model.stride.max()))
            out = lb(frames, swap_rb=True)
            n, h0, w0, _ = frames.shape
            self.letterbox_info = (h0, w0, out.shape[2], out.shape[3])
            return out
        if not isinstance(im, torch.Tensor):
            raise NotImplementedError("sources: a float BCHW tensor in [0,1], or uint8 HWC BGR frames (list of numpy arrays / uint8 NHWC tensor)")
        if im.dim() == 3:
            im = im[None]
        if im.shape[1] != self.model.yaml.get("ch", 3):
            raise ValueError(f"expected {self.model.yaml.get('ch', 3)} input channels, got {im.shape[1]}")
        s = int(self.model.stride.max())
        if im.shape[2] % s or im.shape[3] % s:
            raise ValueError(f"tensor source must have H, W divisible by the model stride {s} (loaders.py:516-584)")
        return im.to(self.device, torch.float32, non_blocking=True).contiguous()

    def _calibrate_fp8(self, im: torch.Tensor) -> None:
        """fp8 storage (BASELINE config 5): one fp16 pass over this batch measures the largest activation the network stores;
        the network-wide activation scale puts it at 224 = half of e4m3's 448 (headroom for other inputs; e4m3 is a floating
        format, so smaller activations keep their 3-bit mantissa down to 2^-9 of the scale).  Weight scales are per output
        channel and need no data (hip_ops.PackedConv)."""
        with H.observe_absmax() as log:
            self.model._predict_once(im, image_dtype=torch.float16)
        amax = float(torch.stack(log).max()) if log else 1.0
        H.set_fp8_act_scale(max(amax / 224.0, 1e-8))
        self.fp8_calibration = {"absmax": amax, "act_scale": H.fp8_act_scale()}

    def _record(self, im: torch.Tensor) -> CompiledForward:
        cf = CompiledForward()
        a = self.args
        if self.dtype == H.FP8 or self.mixed8:
            self._calibrate_fp8(im)
        plan8 = None
        if self.mixed8:
            plan8 = frozenset(a["fp8_layers"]) if a.get("fp8_layers") is not None else self.model.fp8_plan_off_p2()
            self.fp8_calibration["fp8_layers"] = sorted(plan8)
        n, _, h, w = im.shape
        params = torch.tensor(self._box_params(h, w, n), dtype=torch.float32, device=self.device)
        cf.box_params, cf.box_key = params, self._box_params(h, w, n)
        det = self.model.model[-1]
        holder = {}

        def make_bufs(nb, anchors):  # Detect asks for the NMS buffers so that decode can fill the candidate list
            holder["bufs"] = H.NmsBuffers(nb, anchors, int(a["max_det"]), self.device)
            return holder["bufs"]

        det.fused_nms = (make_bufs, float(a["conf"]), self._classes_mask)
        det.fuse_tail = True  # branch tails + decode + filter in one launch where the shape is built
        self.model.fp8_layers = plan8
        try:
            with H.record(cf.plan):
                y, _ = self.model._predict_once(im, image_dtype=self.dtype)
                cf.pred = y
                cf.nms = H.nms(y, float(a["conf"]), float(a["iou"]), max_det=int(a["max_det"]), max_nms=int(a["max_nms"]),
                               max_wh=float(a["max_wh"]), agnostic=bool(a["agnostic_nms"]), nc=self.model.yaml["nc"],
                               classes_mask=self._classes_mask, bufs=holder.get("bufs"), prefiltered="bufs" in holder)
                # construct_result: scale_boxes(img.shape[2:], boxes, orig.shape) — tensor sources are their own
                # original image, so gain 1 / pad 0 and the clip to (h, w) remain (detect/predict.py:59-73)
                H.scale_boxes_(cf.nms, params)
        finally:
            det.fused_nms = None
            det.fuse_tail = False
            self.model.fp8_layers = None
        cf.plan.keep.append(params)
        return cf

    def _box_params(self, h: int, w: int, n: int):
        """Per image (gain, pad_x, pad_y, clip_w, clip_h) of ops.scale_boxes (utils/ops.py:92-127) for the current source: n rows."""
        info = getattr(self, "letterbox_info", None)
        if info is None:
            return [[1.0, 0.0, 0.0, float(w), float(h)]] * n
        rows = []
        for h0, w0, hn, wn in (info if isinstance(info, list) else [info] * n):
            gain = min(hn / h0, wn / w0)
            rows.append([gain, float(round((wn - w0 * gain) / 2 - 0.1)), float(round((hn - h0 * gain) / 2 - 0.1)), float(w0), float(h0)])
        return rows

    def _forward_augment(self, im: torch.Tensor) -> CompiledForward:
        """``augment=True`` (reference predictor.py:306: ``self.model(im, augment=...)`` -> DetectionModel._predict_augment): three passes of the path over
        the image pyramid, merged along the anchors, then the NMS over the merged candidates.  Launched as it comes — nothing recorded or replayed."""
        a = self.args
        if self.dtype == H.FP8 or self.mixed8:
            raise NotImplementedError("augment=True is built for the 16-bit, split-float16 and fp32 storage types")
        cf = CompiledForward()
        y, _ = self.model._predict_augment(im, image_dtype=self.dtype)
        cf.pred = y
        cf.nms = H.nms(y, float(a["conf"]), float(a["iou"]), max_det=int(a["max_det"]), max_nms=int(a["max_nms"]), max_wh=float(a["max_wh"]),
                       agnostic=bool(a["agnostic_nms"]), nc=self.model.yaml["nc"], classes_mask=self._classes_mask)
        n, _, h, w = im.shape
        params = torch.tensor(self._box_params(h, w, n), dtype=torch.float32, device=self.device)
        H.scale_boxes_(cf.nms, params)
        cf.box_params = params
        return cf

    def forward_device(self, im: torch.Tensor) -> CompiledForward:
        """Run one batch; outputs stay on the device in the returned object's ``nms`` buffers."""
        if self.args.get("augment"):
            return self._forward_augment(im)
        key = (tuple(im.shape), self.dtype)
        cf = self._compiled.get(key)
        sig = self.model.weights_signature()
        if cf is not None and cf.weights_sig != sig:
            # the recorded plan (and its hipGraph) holds device pointers to the weight packs as they were at record time;
            # the reference predictor always runs the live model, so re-record after load_state_dict / training / .to()
            self._compiled.pop(key)
            cf = None
        if cf is not None:  # the recorded dy_scale_boxes launch reads this device tensor: refresh it when the source geometry changed
            bp = self._box_params(im.shape[2], im.shape[3], im.shape[0])
            if bp != cf.box_key:
                cf.box_params.copy_(torch.tensor(bp, dtype=torch.float32))
                cf.box_key = bp
        if cf is None:
            while len(self._compiled) >= self.max_compiled:  # a recorded pass owns its buffers and its hipGraph: keep the most recent shapes only
                self._compiled.pop(next(iter(self._compiled)))
            cf = self._compiled[key] = self._record(im)  # recording also executes the launches
            cf.weights_sig = self.model.weights_signature()  # after recording: packing may itself touch caches
            if self.args["graph"]:
                self._capture(cf, im)
            return cf
        if cf.graph is not None:
            if im.data_ptr() == cf.static_in.data_ptr():
                cf.graph.replay()
            elif cf.graph_tail is not None:
                # the caller's own tensor: the launch that reads the image runs from ITS address, the rest is the second graph — no copy of the
                # batch into the graph's static input first (fp32 640 x 640 x 256: 1.26 GB read + written, ~0.6 ms per call)
                cf.plan.rebind_input(im.data_ptr())
                cf.plan.replay(torch.cuda.current_stream().cuda_stream, 0, 1)
                cf.graph_tail.replay()
            else:
                cf.static_in.copy_(im, non_blocking=True)
                cf.graph.replay()
        else:
            cf.plan.rebind_input(im.data_ptr())
            cf.plan.replay(torch.cuda.current_stream().cuda_stream)
        return cf

    def _capture(self, cf: CompiledForward, im: torch.Tensor) -> None:
        """Capture the recorded launches into a hipGraph (replay cost ~10 us instead of ~100 launches)."""
        cf.static_in = im.clone()
        cf.plan.rebind_input(cf.static_in.data_ptr())
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of OTHER host threads (e.g. RCCL's watchdog in a torch.distributed run) must not
        # invalidate this capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            cf.plan.replay(torch.cuda.current_stream().cuda_stream)
        cf.graph = g
        # A second graph without the plan's first launch when that launch is the one reading the image (the fused stem / the layout cast):
        # ``forward_device`` then serves a tensor at ANY address without copying it into ``static_in`` — one eager launch + this graph.
        if cf.plan.input_slot is not None and cf.plan.input_slot[0] == 0 and len(cf.plan.ops) > 1:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                cf.plan.replay(torch.cuda.current_stream().cuda_stream, 1)
            cf.graph_tail = g2

    def profile_layers(self, im: torch.Tensor, iters: int = 3) -> List[dict]:
        """Per-layer device time of the recorded pass for ``im``'s shape — the reference's ``_profile_one_layer`` (nn/tasks.py:171-191: time per layer
        of ``_predict_once(profile=True)``), measured the way this path runs: the plan's launches replayed one by one with HIP events on the
        launch stream and summed by the model layer that issued them (layers folded into a consumer's gather — Upsample, Concat — launch nothing
        and do not appear; NMS and the box rescale come last as 'postprocess').  Returns [{layer, type, launches, ms, kernels}]."""
        cf = self.forward_device(im)
        L, stream = H.lib(), torch.cuda.current_stream().cuda_stream
        tot, names = {}, {}
        for _ in range(iters):
            evs = []
            for i, (fn, args, _) in enumerate(cf.plan.ops):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn(*args, stream)
                e.record()
                evs.append((i, s, e))
                names[i] = (L.dy_last_kernel_name() or b"").decode() or fn.__name__
            torch.cuda.synchronize(self.device)
            for i, s, e in evs:
                tot[i] = tot.get(i, 0.0) + s.elapsed_time(e)
        rows: Dict = {}
        for i in range(len(cf.plan.ops)):
            tag = cf.plan.tags[i] if i < len(cf.plan.tags) and cf.plan.tags[i] is not None else (10 ** 6, "postprocess (NMS, box rescale; an input layout cast where layer 0 has no fused image kernel)")
            r = rows.setdefault(tag, {"layer": tag[0] if tag[0] < 10 ** 6 else None, "type": tag[1], "launches": 0, "ms": 0.0, "kernels": []})
            r["launches"] += 1
            r["ms"] += tot[i] / iters
            if names[i] not in r["kernels"]:
                r["kernels"].append(names[i])
        return [dict(r, ms=round(r["ms"], 4)) for _, r in sorted(rows.items(), key=lambda kv: kv[0][0])]

    def static_input(self, shape) -> Optional[torch.Tensor]:
        cf = self._compiled.get((tuple(shape), self.dtype))
        return None if cf is None else cf.static_in

    def postprocess(self, cf: CompiledForward, im: torch.Tensor, paths=None) -> List[Results]:
        counts = cf.nms.count.tolist()  # the pass's only device->host synchronisation
        names = self.model.names
        out = []
        info = getattr(self, "letterbox_info", None)
        rows = cf.nms.out.clone()  # ONE copy of the padded (N, max_det, 6) rows: the next pass overwrites the buffer, the Results keep views of this one
        for i, k in enumerate(counts):
            one = (info[i] if isinstance(info, list) else info) if info else None
            out.append(Results(im[i], paths[i] if paths else f"image{i}.jpg", names, boxes=rows[i, :k],
                               orig_shape=(one[0], one[1]) if one else im.shape[2:]))
        return out

    # ---- image files and PIL images as sources (reference data/build.py:160-200 check_source / load_inference_source, data/loaders.py:284-420
    # LoadImagesAndVideos, :451-500 LoadPilAndNumpy).  Decoding is host glue in front of the path: PIL here (cv2 is not in the image), so a lossless
    # file (png, bmp, tif) gives the reference's pixels and a jpeg whatever PIL's libjpeg decodes; videos, streams and URLs are not built. ----------
    IMG_FORMATS = {"bmp", "jpeg", "jpg", "mpo", "png", "tif", "tiff", "webp"}

    @classmethod
    def list_image_files(cls, source) -> List[str]:
        """A file, a directory (its `*.*`, sorted), a glob pattern, a `.txt` list of such, or a list of them -> the image files, in the reference's order."""
        import glob
        import os
        from pathlib import Path

        parent = None
        if isinstance(source, (str, Path)) and Path(str(source)).suffix == ".txt":
            parent = Path(str(source)).parent
            source = Path(str(source)).read_text().splitlines()
        files = []
        for q in (sorted(str(v) for v in source) if isinstance(source, (list, tuple)) else [str(source)]):
            a = str(Path(q).absolute())
            if "*" in a:
                files.extend(sorted(glob.glob(a, recursive=True)))
            elif os.path.isdir(a):
                files.extend(sorted(glob.glob(os.path.join(a, "*.*"))))
            elif os.path.isfile(a):
                files.append(a)
            elif parent is not None and (parent / q).is_file():
                files.append(str((parent / q).absolute()))
            else:
                raise FileNotFoundError(f"{q} does not exist")
        images = [f for f in files if f.rpartition(".")[-1].lower() in cls.IMG_FORMATS]
        if len(images) != len(files):
            other = sorted({f.rpartition(".")[-1].lower() for f in files} - cls.IMG_FORMATS)
            if not images:
                raise NotImplementedError(f"no image files in the source (found suffixes {other}); videos / streams are outside the accelerated path")
            LOGGER.warning(f"WARNING skipping {len(files) - len(images)} non-image file(s) ({other}): videos / streams are outside the accelerated path")
        return images

    @staticmethod
    def decode_image(im):
        """A file path or a PIL image -> contiguous HWC BGR uint8 (loaders.py:488-500: RGB, then channels reversed)."""
        import numpy as np
        from PIL import Image

        if not isinstance(im, Image.Image):
            with Image.open(im) as f:
                im = f.convert("RGB")
        elif im.mode != "RGB":
            im = im.convert("RGB")
        return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])

    def _file_pieces(self, files: List[str], batch: int):
        for lo in range(0, len(files), batch):
            names = files[lo : lo + batch]
            yield lo, [self.decode_image(f) for f in names], names

    # ---- a source larger than one batch (reference predictor.py:221-298, stream_inference: `for self.batch in self.dataset`) -------------------
    def _chunks(self, source, batch: int):
        """``source`` cut into pieces of ``batch`` images: a BCHW float tensor, a uint8 (N, H, W, 3) tensor, or a list of HWC frames."""
        n = len(source)
        for lo in range(0, n, batch):
            yield lo, source[lo : lo + batch], None

    def stream_batches(self, source, batch: int, pieces=None):
        """Generator over the images of ``source``, run ``batch`` at a time, one ``Results`` per image in order.  The device works one batch
        ahead of the host: batch k + 1 is preprocessed and enqueued (its launches, a device copy of batch k's output rows in front of them, an
        asynchronous copy of the kept counts to pinned memory) before batch k's ``Results`` are built, so the host side of postprocess — the
        one synchronisation and a Python object per image — runs under the next batch's kernels.  What the reference's loop does batch by batch
        (predictor.py:246-298), without its per-batch device synchronisations."""
        names = self.model.names
        pending = None

        def finish(item):
            lo, im, rows, counts_host, ev, info, t_pre, t_inf, paths, frames = item
            t0 = time.perf_counter()
            ev.synchronize()
            counts = counts_host.tolist()
            out = []
            for i, k in enumerate(counts):
                one = (info[i] if isinstance(info, list) else info) if info else None
                r = Results(frames[i] if frames is not None else im[i], paths[i] if paths else f"image{lo + i}.jpg", names, boxes=rows[i, :k],
                            orig_shape=(one[0], one[1]) if one else im.shape[2:])
                out.append(r)
            dt = (time.perf_counter() - t0) * 1e3 / max(len(out), 1)
            for r in out:  # host-side times per image (the device runs ahead: no synchronisation is placed around the stages)
                r.speed = {"preprocess": t_pre, "inference": t_inf, "postprocess": dt}
            return out

        for lo, piece, paths in (pieces if pieces is not None else self._chunks(source, batch)):
            t0 = time.perf_counter()
            im = self.preprocess(piece)
            t1 = time.perf_counter()
            cf = self.forward_device(im)
            rows = cf.nms.out.clone()  # stream-ordered behind this batch's launches, in front of the next batch's (which overwrite the buffer)
            counts_host = torch.empty(cf.nms.count.shape, dtype=cf.nms.count.dtype, pin_memory=True)
            counts_host.copy_(cf.nms.count, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            t2 = time.perf_counter()
            n = max(im.shape[0], 1)
            item = (lo, im, rows, counts_host, ev, getattr(self, "letterbox_info", None), (t1 - t0) * 1e3 / n, (t2 - t1) * 1e3 / n, paths,
                    piece if paths is not None else None)  # (file sources keep the decoded frame as the result's orig_img)
            if pending is not None:
                yield from finish(pending)
            pending = item
        if pending is not None:
            yield from finish(pending)

    def __call__(self, source, stream: bool = False):
        from pathlib import Path

        is_path = lambda v: isinstance(v, (str, Path))  # noqa: E731
        if is_path(source) or (isinstance(source, (list, tuple)) and len(source) > 0 and all(is_path(v) for v in source)):
            # image files: the reference's loader hands over `batch` files at a time (default 1: every image letterboxed to ITS minimum rectangle)
            files = self.list_image_files(source)
            gen = self.stream_batches(None, 0, pieces=self._file_pieces(files, int(self.args.get("batch") or 1)))
            return gen if stream else list(gen)
        if type(source).__module__.startswith("PIL.") or (isinstance(source, (list, tuple)) and len(source) > 0 and type(source[0]).__module__.startswith("PIL.")):
            ims = list(source) if isinstance(source, (list, tuple)) else [source]  # LoadPilAndNumpy: ONE batch of all of them
            paths = [getattr(im, "filename", "") or f"image{i}.jpg" for i, im in enumerate(ims)]
            gen = self.stream_batches(None, 0, pieces=iter([(0, [self.decode_image(im) for im in ims], paths)]))
            return gen if stream else list(gen)
        batch = self.args.get("batch")
        many = (isinstance(source, (list, tuple)) and len(source) > 0 and getattr(source[0], "ndim", 0) == 3) or (isinstance(source, torch.Tensor) and source.dim() == 4)
        if batch and many and len(source) > int(batch):  # more images than one batch: the reference's dataset loop
            gen = self.stream_batches(source, int(batch))
            return gen if stream else list(gen)
        prof = [ops.Profile(device=self.device) for _ in range(3)]
        with prof[0]:
            im = self.preprocess(source)
        with prof[1]:
            cf = self.forward_device(im)
        with prof[2]:
            results = self.postprocess(cf, im)
        n = max(len(results), 1)
        for r in results:
            r.speed = {"preprocess": prof[0].dt * 1e3 / n, "inference": prof[1].dt * 1e3 / n,
                       "postprocess": prof[2].dt * 1e3 / n}
        if self.args["verbose"]:
            LOGGER.info(f"{len(results)} images: " + ", ".join(f"{len(r)} boxes" for r in results))
        return iter(results) if stream else results
