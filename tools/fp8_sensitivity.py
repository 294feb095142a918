"""Which parts of the network tolerate fp8 (e4m3) storage / arithmetic?  GPU box tool (not product code).

The fp16 pass is run layer by layer with FAKE quantisation: a convolution whose output buffer a plan stores in fp8 has its output
rounded to the e4m3 grid (network-wide activation scale, as engine/predictor.py::_calibrate_fp8 sets it), a convolution whose inputs
are all fp8 buffers has its folded weights rounded to e4m3 with one scale per output channel (hip_ops.PackedConv's fp8 recipe).  fp16
represents every e4m3 value exactly and the fp16 kernels accumulate in fp32, so the result equals what fp8 operands in an fp8 MFMA
give, up to summation order — without needing an fp8 kernel for every shape.  Every plan is scored against the REAL reference's rows
(tests/golden/big.npz) with the bench's own gate (utils/parity.py::detection_parity).

    python tools/fp8_sensitivity.py [s640bench|x1536] > gpurun_out/fp8_sensitivity_<tag>.jsonl
"""
import json
import os
import sys

os.environ.setdefault("DYOLO_L2E", "0")  # reference units: the e4m3 grid is scale free, the log2(e) domain changes nothing here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import yaml  # noqa: E402

import bench  # noqa: E402
import drone_yolo_amd as D  # noqa: E402
from drone_yolo_amd import hip_ops as H  # noqa: E402
from drone_yolo_amd.engine.predictor import DetectionPredictor  # noqa: E402
from drone_yolo_amd.nn.modules import C2f, Conv, Detect, RepVGGBlock, SPPF  # noqa: E402
from drone_yolo_amd.nn.modules.conv import PlainConv2d  # noqa: E402
from drone_yolo_amd.utils import parity as PR  # noqa: E402

FP8 = torch.float8_e4m3fn
STATE = {"cur": None, "wq": set(), "oq": set(), "scale": 1.0}


def fq_act_(t: torch.Tensor) -> None:
    s = STATE["scale"]
    q = (t.float() / s).clamp_(-448.0, 448.0).to(FP8).float() * s
    t.copy_(q.to(t.dtype))


def fq_weight(w: torch.Tensor) -> torch.Tensor:
    w = w.detach().float()
    ws = (w.flatten(1).abs().amax(1) / 448.0).clamp_min(1e-12).view(-1, 1, 1, 1)
    return (w / ws).to(FP8).float() * ws


_conv2d, _fold = H.conv2d, H.domain_fold


def conv2d(x, pc, out=None, **kw):
    y = _conv2d(x, pc, out=out, **kw)
    if STATE["cur"] in STATE["oq"] and not kw.get("out_f32"):
        fq_act_(y)
    return y


def domain_fold(w, b, silu, raw_input=False, raw_output=False):
    w, b, act = _fold(w, b, silu, raw_input=raw_input, raw_output=raw_output)
    if STATE["cur"] in STATE["wq"]:
        w = fq_weight(w)
    return w, b, act


H.conv2d, H.domain_fold = conv2d, domain_fold


def build(tag):
    meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", tag)
    d = yaml.safe_load(open(os.path.join(ROOT, "drone-yolo_amd", "cfg", "models", "v8", meta["yaml"])))
    d["scale"], d["nc"] = meta["scale"], meta["nc"]
    d["yaml_file"] = meta["yaml"].replace("yolov8", f"yolov8{meta['scale']}")
    model = D.DetectionModel(d, nc=meta["nc"], verbose=False)
    model.load_state_dict(bench.fixture_weights(model, meta))
    model.fuse_stem2 = False
    model.fuse_stem = False  # layer 0 through Conv.__call__ (the hook below must see it); the image arrives as an NHWC view padded to one chunk
    for m in model.modules():
        if isinstance(m, C2f):
            m.fuse_block = False
        if isinstance(m, Detect):
            m.fuse_tail = m.fuse_first = m.fuse_branch = False
    return model, x, exp_rows, exp_idx


def conv_modules(model):
    """[(module, top-level layer index, is the layer's first conv, FLOP weight, role)]"""
    out = []
    for layer in model.model:
        first = None
        if isinstance(layer, (Conv, RepVGGBlock)):
            first = {layer}
        elif isinstance(layer, (C2f, SPPF)):
            first = {layer.cv1}
        elif isinstance(layer, Detect):
            first = {seq[0] for seq in list(layer.cv2) + list(layer.cv3)}
        for name, m in layer.named_modules():
            if isinstance(m, (Conv, RepVGGBlock, PlainConv2d)):
                role = "bottleneck" if ".m." in f".{name}." or name.startswith("m.") else ("detect" if isinstance(layer, Detect) else "other")
                out.append((m, layer.i, m in (first or ()), role, name))
    return out


def run(model, x, exp_rows, exp_idx, wq, oq, label, extra=None):
    STATE["wq"], STATE["oq"] = set(wq), set(oq)
    model.drop_packed()
    pred = DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="fp16", device=0))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    par = PR.detection_parity(cf.nms, exp_rows, exp_idx, conf=0.25, margin=5e-4)
    rec = {"plan": label, "n_wq": len(wq), "n_oq": len(oq), **{k: par[k] for k in ("match_rate", "missed", "extra", "iou_min", "iou_mean", "ref_detections", "kept_detections")},
           **(extra or {})}
    print(json.dumps(rec), flush=True)
    del pred, cf
    return rec


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "s640bench"
    model, x, exp_rows, exp_idx = build(tag)
    mods = conv_modules(model)
    for m, *_ in mods:
        m.register_forward_pre_hook(lambda mod, args, kwargs=None: STATE.__setitem__("cur", mod))
    # activation scale: largest stored activation of an fp16 pass -> 224 (engine/predictor.py::_calibrate_fp8)
    STATE["wq"], STATE["oq"] = set(), set()
    p0 = DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="fp16", device=0))
    with H.observe_absmax() as log:
        p0.forward_device(p0.preprocess(x))
    torch.cuda.synchronize()
    STATE["scale"] = max(float(torch.stack(log).max()) / 224.0, 1e-8)
    print(json.dumps({"fixture": tag, "act_scale": STATE["scale"], "convs": len(mods)}), flush=True)
    del p0
    layers = sorted({li for _, li, *_ in mods})
    det_i = max(layers)

    def plan(layer_set, detect_first=False, bottleneck_only=False):
        """fp8 buffers: every conv output inside the layers of ``layer_set``; fp8 arithmetic (weights rounded): every conv all of
        whose inputs are fp8 buffers = the non-first convs of those layers, and a layer's first conv when all its sources are in the set."""
        wq, oq = set(), set()
        srcs = model._srcs if hasattr(model, "_srcs") else None
        for m, li, is_first, role, name in mods:
            if bottleneck_only:
                in_c2f = isinstance(model.model[li], C2f) and li in layer_set
                if in_c2f and (role == "bottleneck" or is_first):
                    oq.add(m)  # y0 | y1 and every Bottleneck output are fp8; the block's output (cv2) stays fp16
                if in_c2f and not is_first:
                    wq.add(m)
                continue
            if li in layer_set and not isinstance(m, PlainConv2d):
                oq.add(m)
            fed8 = all(s in layer_set for s in _conv_sources(model, li)) if is_first else (li in layer_set)
            if li == det_i and li not in layer_set:
                fed8 = is_first and detect_first and all(s in layer_set for s in _conv_sources(model, li, m))
            if fed8:
                wq.add(m)
        return wq, oq

    all_l = [l for l in layers if l != det_i]
    run(model, x, exp_rows, exp_idx, set(), set(), "fp16 everywhere (sanity)")
    wq, oq = plan(set(all_l) | {det_i})
    run(model, x, exp_rows, exp_idx, wq, oq, "fp8 everywhere (what --dtype fp8 stores)")
    wq, oq = plan(set(all_l), detect_first=True)
    run(model, x, exp_rows, exp_idx, wq, oq, "trunk fp8, Detect reads fp8 / computes fp16 from its first convs' outputs on")
    wq, oq = plan(set(all_l), bottleneck_only=True)
    run(model, x, exp_rows, exp_idx, wq, oq, "C2f internals fp8 (cv1 out, Bottlenecks, cv2 arithmetic), block boundaries fp16")
    # every layer that does not feed the highest-resolution Detect level (the P2 map, where most detections of these models live)
    top = _conv_sources(model, det_i)
    p2_src = model._srcs[det_i][0]
    anc, todo = set(), [p2_src]
    while todo:
        t = todo.pop()
        if t in anc or t < 0:
            continue
        anc.add(t)
        todo += [q for q in model._srcs[t] if q >= 0 and t > 0]
    deep = [l for l in all_l if l not in anc]
    wq, oq = plan(set(deep), detect_first=True)
    run(model, x, exp_rows, exp_idx, wq, oq, f"layers off the P2 path fp8 ({min(deep)}..{max(deep)}), Detect levels 1.. read fp8 (first convs), everything else fp16", {"layers": deep})
    if "--quick" in sys.argv:
        return
    bb = [l for l in all_l if l <= 9]
    nk = [l for l in all_l if l > 9]
    for lab, ls in (("backbone (layers 0-9) fp8", bb), ("neck (layers 10-27) fp8", nk)):
        wq, oq = plan(set(ls))
        run(model, x, exp_rows, exp_idx, wq, oq, lab)
    for l in all_l + [det_i]:  # one layer at a time
        wq, oq = plan({l})
        run(model, x, exp_rows, exp_idx, wq, oq, f"only layer {l} ({type(model.model[l]).__name__}) fp8", {"layer": l})
    for l in all_l:  # leave one out
        wq, oq = plan(set(all_l) - {l}, detect_first=True)
        run(model, x, exp_rows, exp_idx, wq, oq, f"trunk fp8 except layer {l} ({type(model.model[l]).__name__})", {"left_out": l})


def _conv_sources(model, li, m=None):
    """Layers whose output buffers the first conv(s) of layer ``li`` read (Upsample / Concat resolved to their producers)."""
    from drone_yolo_amd.nn.modules import Concat, Upsample

    if not hasattr(model, "_srcs"):
        model._plan_graph()
    out, todo = [], list(model._srcs[li])
    if m is not None and isinstance(model.model[li], Detect):  # one Detect level reads one source
        det = model.model[li]
        for i in range(det.nl):
            if m in (det.cv2[i][0], det.cv3[i][0]):
                todo = [model._srcs[li][i]]
    while todo:
        s = todo.pop()
        if isinstance(model.model[s], (Concat, Upsample)):
            todo += list(model._srcs[s])
        else:
            out.append(s)
    return out


if __name__ == "__main__":
    main()
