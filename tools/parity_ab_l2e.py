"""A/B of the log2(e)-scaled activation domain (DY_ACT_SILU_L2E) on parity: fp16 storage with and without it on every full-size
fixture of tests/golden/big.npz (reference PyTorch-CPU fp32 rows).  GPU box tool:  python tools/parity_ab_l2e.py > gpurun_out/parity_ab_l2e.jsonl
VERDICT r3 item 6: if the scaled domain loses detections on three of four fixtures it is not a lottery."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import yaml  # noqa: E402

import bench  # noqa: E402
import drone_yolo_amd as D  # noqa: E402
from drone_yolo_amd.engine.predictor import DetectionPredictor  # noqa: E402
from drone_yolo_amd.utils import parity as PR  # noqa: E402

KEYS = ("ref_detections", "kept_detections", "match_rate", "missed", "extra", "missed_clear", "extra_clear", "iou_min", "iou_mean", "counts_equal", "kept_sets_identical")


def build(meta):
    d = yaml.safe_load(open(os.path.join(ROOT, "drone-yolo_amd", "cfg", "models", "v8", meta["yaml"])))
    d["scale"], d["nc"] = meta["scale"], meta["nc"]
    d["yaml_file"] = meta["yaml"].replace("yolov8", f"yolov8{meta['scale']}")
    model = D.DetectionModel(d, nc=meta["nc"], verbose=False)
    model.load_state_dict(bench.fixture_weights(model, meta))
    return model


for tag in ("s640bench", "s640b4", "x1536", "l1280t8"):
    for l2e in ("1", "0"):
        os.environ["DYOLO_L2E"] = l2e
        meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", tag)
        model = build(meta)
        if tag == "l1280t8":
            par = bench.tiled_record(model, "fp16", 0, frames=1)["parity_per_tile"]
        else:
            pred = DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="fp16", device=0))
            cf = pred.forward_device(pred.preprocess(x))
            torch.cuda.synchronize()
            par = PR.detection_parity(cf.nms, exp_rows, exp_idx, conf=0.25, margin=5e-4)
            del pred, cf
        print(json.dumps({"test": "A/B scaled activation domain", "fixture": tag, "dtype": "fp16", "DY_ACT_SILU_L2E": l2e == "1", **{k: par[k] for k in KEYS if k in par}}), flush=True)
        del model
        torch.cuda.empty_cache()
