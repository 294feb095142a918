"""ORACLE tooling — checks that a checkpoint written by this package (nn/checkpoint.py::save_reference_checkpoint, the
``last.pt`` of ``YOLO.train``) is restored by the REAL reference as its own classes and computes the same output.

Build container only (imports /root/reference through oracle/make_golden.py::import_reference):

    python oracle/check_checkpoint_roundtrip.py [path.pt]

Without a path it writes one from a seeded Drone-YOLO-n first.  Prints "roundtrip ok" and exits 0 on success.
"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import drone_yolo_amd as D
    from drone_yolo_amd.nn.checkpoint import save_reference_checkpoint
    from oracle import drone_yolo_oracle as O
    from oracle.make_golden import import_reference, our_yaml

    path = sys.argv[1] if len(sys.argv) > 1 else None
    d = our_yaml("yolov8-p2-repvgg.yaml", "n", 10)
    if path is None:
        model = D.DetectionModel(dict(d), nc=10, verbose=False)
        sd = O.seeded_state_dict(model.state_dict(), 31, cls_bias=-1.5)
        model.load_state_dict(sd)
        path = os.path.join(tempfile.mkdtemp(), "last.pt")
        opt = {"state": {0: {"momentum_buffer": torch.ones(3)}}, "param_groups": [{"lr": 0.01, "params": [0]}]}
        save_reference_checkpoint(path, model, model.state_dict(), extra={"epoch": 4, "updates": 7, "optimizer": opt, "train_args": {"imgsz": 64, "batch": 2}})
    R = import_reference()
    ck = torch.load(path, map_location="cpu", weights_only=False)  # the reference's own loader does exactly this (tasks.py:805-835)
    assert set(("epoch", "best_fitness", "model", "ema", "updates", "optimizer", "train_args", "date", "version")) <= set(ck), sorted(ck)
    ema = ck["ema"]
    assert type(ema) is R.tasks.DetectionModel, type(ema)
    kinds = {type(m).__module__ + "." + type(m).__name__ for m in ema.modules()}
    assert not any(k.startswith("drone_yolo_amd") for k in kinds), kinds
    assert "ultralytics.nn.modules.block.RepVGGBlock" in kinds and "ultralytics.nn.modules.head.Detect" in kinds
    assert next(ema.parameters()).dtype == torch.float16
    model = ema.float().eval()  # attempt_load_one_weight: (ckpt.get("ema") or ckpt["model"]).float() (tasks.py:906)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        y_ref, _ = model(x)
        sd = {k: v.float() if v.is_floating_point() else v for k, v in model.state_dict().items()}
        y_or, _ = O.forward(dict(ema.yaml), sd, x, fused=False)
    err = float((y_ref - y_or).abs().max())
    assert err < 1e-3, err
    if isinstance(ck["optimizer"], dict) and ck["optimizer"].get("state"):
        st = next(iter(ck["optimizer"]["state"].values()))
        assert all(v.dtype == torch.float16 for k, v in st.items() if k != "step" and isinstance(v, torch.Tensor) and v.is_floating_point())
    print(f"roundtrip ok: {path} restored by the reference as {type(ema).__module__}.{type(ema).__name__}, {len(kinds)} module kinds, "
          f"max |y_reference - y_oracle| = {err:.2e}")


if __name__ == "__main__":
    main()
