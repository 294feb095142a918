"""GPU parity of the module graph and the end-to-end predict path against the oracle and the golden
vectors captured from the real reference.

Tolerance (BASELINE.json north_star): class / kept-index bit-exact and box IoU >= 0.999 against the
fp32 CPU path.  That bar is asserted for the fp32 device path.  For bf16 / fp16 storage (the
throughput modes; every layer boundary rounds activations to 8 / 11 mantissa bits) the test asserts
what the arithmetic can deliver and prints the measured numbers: matched-detection rate and IoU.
"""
import json
import os

import numpy as np
import pytest
import torch

import drone_yolo_amd as D
from drone_yolo_amd import hip_ops as H
from drone_yolo_amd.nn import modules as M
from drone_yolo_amd.nn.tasks import initialize_weights
from oracle import drone_yolo_oracle as O
from tests._util import ROOT, box_iou_pairs, golden, load_yaml, meta, quantize, split_rows

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
IDS = ["f32", "bf16", "f16"]
RTOL = {torch.float32: 1e-4, torch.bfloat16: 4e-2, torch.float16: 6e-3}


def _report(name, payload):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **payload}) + "\n")


def _load_seeded(mod, seed, device):
    sd = O.seeded_state_dict(mod.state_dict(), seed)
    mod.load_state_dict(sd)
    initialize_weights(mod)
    return mod.to(device).eval()


def _dev(t, dtype, device):
    return t.permute(0, 2, 3, 1).contiguous().to(device, dtype).permute(0, 3, 1, 2)


def _close(got, ref, dtype, what):
    scale = float(ref.abs().max())
    err = float((got.float().cpu() - ref).abs().max())
    _report(what, {"dtype": str(dtype), "max_abs_err": err, "scale": scale})
    assert err <= RTOL[dtype] * scale, f"{what} [{dtype}]: max|err| {err:.4e}, scale {scale:.3f}"


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_modules_match_reference_vectors(dtype, device):
    """Conv / DWConv / RepVGGBlock / Bottleneck / C2f / SPPF / Detect against outputs of the REAL reference modules."""
    g = golden("per_op.npz")
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    c1, c2, k, s = (int(v) for v in g["conv_args"])
    _close(_load_seeded(M.Conv(c1, c2, k, s), int(g["conv_seed"]), device)(_dev(t("conv_x"), dtype, device)), t("conv_y"), dtype, "Conv k3 s2")
    c1, c2, k, s = (int(v) for v in g["conv1_args"])
    _close(_load_seeded(M.Conv(c1, c2, k, s), int(g["conv1_seed"]), device)(_dev(t("conv1_x"), dtype, device)), t("conv1_y"), dtype, "Conv k1")
    c1, c2, k, s = (int(v) for v in g["dw_args"])
    _close(_load_seeded(M.DWConv(c1, c2, k, s), int(g["dw_seed"]), device)(_dev(t("dw_x"), dtype, device)), t("dw_y"), dtype, "DWConv")
    for tag in ("rep_s2", "rep_id"):
        c1, c2, k, s = (int(v) for v in g[f"{tag}_args"])
        m = _load_seeded(M.RepVGGBlock(c1, c2, 3, s), int(g[f"{tag}_seed"]), device)
        _close(m(_dev(t(f"{tag}_x"), dtype, device)), t(f"{tag}_y"), dtype, f"RepVGGBlock {tag}")
    m = _load_seeded(M.Bottleneck(16, 16, True, 1, k=((3, 3), (3, 3)), e=1.0), int(g["bott_seed"]), device)
    _close(m(_dev(t("bott_x"), dtype, device)), t("bott_y"), dtype, "Bottleneck")
    for tag in ("c2f_a", "c2f_b"):
        c1, c2, n, sc = (int(v) for v in g[f"{tag}_args"])
        m = _load_seeded(M.C2f(c1, c2, n, bool(sc)), int(g[f"{tag}_seed"]), device)
        _close(m(_dev(t(f"{tag}_x"), dtype, device)), t(f"{tag}_y"), dtype, f"C2f {tag}")
    m = _load_seeded(M.SPPF(32, 32, 5), int(g["sppf_seed"]), device)
    _close(m(_dev(t("sppf_x"), dtype, device)), t("sppf_y"), dtype, "SPPF")
    M.Detect.legacy = True
    det = M.Detect(nc=5, ch=(16, 32))
    det = _load_seeded(det, int(g["det_seed"]), device)
    det.stride = torch.tensor([8.0, 16.0])
    y, raw = det([_dev(t("det_x0"), dtype, device), _dev(t("det_x1"), dtype, device)])
    _close(raw[0], t("det_raw0"), dtype, "Detect raw0")
    _close(raw[1], t("det_raw1"), dtype, "Detect raw1")
    assert tuple(y.shape) == tuple(t("det_y").shape)
    if dtype == torch.float32:
        assert torch.allclose(y.cpu(), t("det_y"), rtol=1e-4, atol=5e-3)


def _build(tag, g, device):
    m = meta(g, tag)
    d = load_yaml(m["yaml"], m["scale"], m["nc"])
    model = D.DetectionModel(dict(d), nc=m["nc"], verbose=False)
    sd = O.seeded_state_dict(model.state_dict(), m["seed"], cls_bias=m["cls_bias"])
    model.load_state_dict(sd)
    b, h, w = m["shape"]
    x = torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(m["seed"]))
    return m, d, sd, model, x


def _match_stats(rows, idx, exp_rows, exp_idx):
    """Per image: fraction of reference detections reproduced (same anchor index AND class) and their IoU."""
    exp_map = {int(a): r for a, r in zip(exp_idx, exp_rows)}
    got_map = {int(a): r for a, r in zip(idx, rows)}
    common = [a for a in exp_map if a in got_map and int(got_map[a][5]) == int(exp_map[a][5])]
    if not exp_map:
        return 1.0, 1.0, 1.0
    ious = box_iou_pairs(np.stack([got_map[a][:4] for a in common]), np.stack([exp_map[a][:4] for a in common])) if common else np.zeros(1)
    return len(common) / len(exp_map), float(ious.min()), float(ious.mean())


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("tag", ["n64", "n128", "sf_n64", "v8n320", "s640"])
def test_end_to_end_against_reference_vectors(tag, dtype, device):
    g = golden("e2e.npz")
    m, d, sd, model, x = _build(tag, g, device)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype=dtype, device=0))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    y = cf.pred.cpu()
    if f"{tag}__y" in g.files:
        yref = torch.from_numpy(g[f"{tag}__y"])
    else:
        yref, y = torch.from_numpy(g[f"{tag}__y_sub"]), y[:, :, ::37]
    box_err = float((y[:, :4] - yref[:, :4]).abs().max())
    cls_err = float((y[:, 4:] - yref[:, 4:]).abs().max())
    counts = cf.nms.count.cpu().tolist()
    # golden rows are the reference's non_max_suppression output; the predictor additionally applies
    # construct_result's scale_boxes/clip_boxes to the input size (detect/predict.py:59-73) -> clip the expectation
    exp_rows = [np.concatenate((O.clip_boxes(torch.from_numpy(r[:, :4].copy()), x.shape[2:]).numpy(), r[:, 4:]), 1) if len(r) else r
                for r in split_rows(g[f"{tag}__det"], g[f"{tag}__n"])]
    exp_idx = split_rows(g[f"{tag}__det_idx"], g[f"{tag}__n"])
    stats = []
    for i, c in enumerate(counts):
        rows, idx = cf.nms.out[i, :c].cpu().numpy(), cf.nms.index[i, :c].cpu().numpy()
        stats.append(_match_stats(rows, idx, exp_rows[i], exp_idx[i]))
    match = min(s[0] for s in stats)
    iou_min = min(s[1] for s in stats)
    _report(f"e2e {tag}", {"dtype": str(dtype), "box_max_err_px": box_err, "cls_max_err": cls_err, "counts": counts,
                           "ref_counts": [int(v) for v in g[f"{tag}__n"]], "match_rate_min": match, "iou_min": iou_min,
                           "iou_mean": float(np.mean([s[2] for s in stats]))})
    if dtype == torch.float32:
        # the north-star bar: kept set, order, classes identical; IoU >= 0.999; raw outputs to fp32 round-off
        assert box_err < 2e-2 and cls_err < 1e-4, (box_err, cls_err)
        assert counts == [int(v) for v in g[f"{tag}__n"]]
        for i, c in enumerate(counts):
            got_idx, got_cls = cf.nms.index[i, :c].cpu().numpy(), cf.nms.out[i, :c, 5].cpu().numpy()
            # identical kept SET and identical class per kept anchor (bit-exact integer outputs) ...
            assert sorted(got_idx.tolist()) == sorted(exp_idx[i].tolist()), f"{tag} image {i}: kept anchor sets differ"
            cls_of = {int(a): int(k) for a, k in zip(exp_idx[i], exp_rows[i][:, 5])}
            assert all(cls_of[int(a)] == int(k) for a, k in zip(got_idx, got_cls)), f"{tag} image {i}: classes differ"
            # ... in identical order, except that two detections whose reference scores are closer than fp32
            # round-off of the network (a few 1e-7) may swap places
            score_of = {int(a): float(sc) for a, sc in zip(exp_idx[i], exp_rows[i][:, 4])}
            for k in np.nonzero(got_idx != exp_idx[i])[0]:
                assert abs(score_of[int(got_idx[k])] - float(exp_rows[i][k, 4])) < 2e-6, f"{tag} image {i}: order differs at rank {k}"
        assert iou_min >= 0.999, iou_min
    else:
        # reduced-precision storage: a score within rounding of conf or of a neighbour may flip.  Floors = the measured level
        # (profiles/r01_parity_report.jsonl, r02) minus a margin: bf16 may lose 3 % of the reference detections (at least one:
        # the small cases keep 2..17 boxes), fp16 1 %; matched boxes IoU >= 0.998 (bf16) / 0.9995 (fp16)
        # two-sided: detections the reference does not keep ("extra") are bounded like the ones it keeps and we lose ("missed").
        # A detection whose score sits within the storage type's score error of `conf` is decided by rounding (the v8n320 case keeps 14 boxes,
        # ALL scored 0.2500 .. 0.2557): such flips are not counted; everything else is, up to `allowed` (NMS near-ties).
        # (r05: the bf16 floor is 0.998 again; bf16 inference packs of 128-channel layers run the kernel they ran when the floor was set.)
        tol, iou_floor = (0.03, 0.998) if dtype == torch.bfloat16 else (0.01, 0.9995)
        margin = 2e-3 if dtype == torch.bfloat16 else 5e-4
        for i in range(len(counts)):
            n_ref = max(len(exp_idx[i]), 1)
            allowed = max(1, int(tol * n_ref))
            got_rows = cf.nms.out[i, :counts[i]].cpu().numpy()
            ref_keys = {(int(a), int(r[5])): float(r[4]) for a, r in zip(exp_idx[i], exp_rows[i])}
            got_keys = {(int(a), int(r[5])): float(r[4]) for a, r in zip(cf.nms.index[i, :counts[i]].cpu().tolist(), got_rows)}
            missed = [k for k, sc in ref_keys.items() if k not in got_keys and sc > 0.25 + margin]
            extra = [k for k, sc in got_keys.items() if k not in ref_keys and sc > 0.25 + margin]
            assert len(missed) <= allowed, f"{tag} [{dtype}] image {i}: {len(missed)} of {n_ref} reference detections (scored beyond conf + {margin}) lost"
            assert len(extra) <= allowed, f"{tag} [{dtype}] image {i}: {len(extra)} detections (scored beyond conf + {margin}) the reference does not keep (of {n_ref})"
            # ... and a loose bound on ALL flips, near-threshold ones included: a regression that drops every detection scored just above
            # conf (v8n320: all 14) must not pass because each one is "decided by rounding"
            tot_missed = sum(1 for k in ref_keys if k not in got_keys)
            tot_extra = sum(1 for k in got_keys if k not in ref_keys)
            loose = max(2, int(0.15 * n_ref))
            assert tot_missed <= loose and tot_extra <= loose, f"{tag} [{dtype}] image {i}: {tot_missed} missed / {tot_extra} extra of {n_ref} in total (bound {loose})"
        assert iou_min >= iou_floor, f"{tag} [{dtype}]: min IoU {iou_min:.5f} < {iou_floor}"


def _bench_model(meta, device):
    """The model bench.py times: Drone-YOLO of the fixture's scale with bench.synthetic_state_dict(seed 0) weights."""
    import bench

    d = load_yaml(meta["yaml"], meta["scale"], meta["nc"])
    d["yaml_file"] = meta["yaml"].replace("yolov8", f"yolov8{meta['scale']}")
    model = D.DetectionModel(dict(d), nc=meta["nc"], verbose=False)
    model.load_state_dict(bench.fixture_weights(model, meta))
    return model


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("tag", ["s640bench", "s640b4", "s640b4lo"])
def test_bench_configuration_against_reference_rows(tag, dtype, device):
    """BASELINE config 2 (Drone-YOLO-s, 4 images of 640x640) against the rows the REAL reference computed on CPU in fp32
    (tests/golden/big.npz).  s640bench — bench.py's own weights and input recipe — is the gate bench.py prints as `parity`; s640b4
    carries the e2e golden's weights.  The bar (IoU >= 0.999, class / index exact up to 1 % of near-tie flips) is asserted for
    fp32 (exact) and for the headline dtype fp16; bf16 is held to its measured level."""
    from drone_yolo_amd.utils import parity as PR

    meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", tag)
    model = _bench_model(meta, device)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype=dtype, device=0))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    g = golden("big.npz")
    y_sub = cf.pred[:, :, ::199].cpu()
    box_err = float((y_sub[:, :4] - torch.from_numpy(g[f"{tag}__y_sub"])[:, :4]).abs().max())
    cls_err = float((y_sub[:, 4:] - torch.from_numpy(g[f"{tag}__y_sub"])[:, 4:]).abs().max())
    # score error of the storage type at this depth (measured r02 / r03: fp16 3e-4 .. 4e-4, bf16 1.3e-3 .. 2.9e-3): a detection scored that
    # close to `conf` is kept or dropped by rounding alone
    margin = {torch.float32: 0.0, torch.float16: 5e-4, torch.bfloat16: 4e-3}[dtype]
    par = PR.detection_parity(cf.nms, exp_rows, exp_idx, conf=0.25, margin=margin)
    _report(f"bench-config {tag}", {"dtype": str(dtype), "box_max_err_px": box_err, "cls_max_err": cls_err, **par})
    if dtype == torch.float32:
        assert box_err < 2e-2 and cls_err < 1e-4, (box_err, cls_err)
        assert par["counts_equal"] and par["kept_sets_identical"] and par["match_rate"] == 1.0 and par["iou_min"] >= 0.999, par
    else:
        # r03: the gate is two-sided — `missed` (reference detections lost) and `extra` (kept here, absent in the reference) are bounded
        # separately.  fp16 (the headline dtype): the IoU bar holds; at most 0.6 % each way in all (measured r03: 0.18 % .. 0.36 % missed,
        # 0.09 % extra — ~1,100 detections whose scores have a density of ~7,500 per unit near conf, times a score error of 3e-4, puts 2 - 5
        # of them at the mercy of rounding), and at most 0.3 % each way among those scored CLEAR of the threshold (NMS near-ties).
        # bf16 (not a headline dtype): 4 % each way, the worst single box at IoU >= 0.993.
        tol_all, tol_clear, iou_floor = (0.006, 0.003, 0.999) if dtype == torch.float16 else (0.04, 0.03, 0.993)
        if tag == "s640b4lo" and dtype == torch.bfloat16:
            tol_all = tol_clear = 0.20  # every one of this case's 45 detections scores within 0.08 logit of conf: bf16 scores (+-2e-3) flip 7 of them (r02)
        n = par["ref_detections"]
        assert par["missed"] <= max(1, int(tol_all * n)) and par["extra"] <= max(1, int(tol_all * n)), par
        assert par["missed_clear"] <= max(1, int(tol_clear * n)) and par["extra_clear"] <= max(1, int(tol_clear * n)), par
        assert par["iou_min"] >= iou_floor, par


def test_plan_follows_the_live_weights(device):
    """a recorded LaunchPlan / hipGraph bakes in pointers to the weight packs.  After load_state_dict (and after a
    train() <-> eval() round trip) the predictor must re-record instead of replaying stale weights: its output equals a
    fresh predictor's, as the reference predictor always runs the live model."""
    g = golden("e2e.npz")
    m, d, sd, model, x = _build("n128", g, device)
    x = x.to(device)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0, graph=True))
    y0 = pred.forward_device(x).pred.clone()
    assert torch.equal(pred.forward_device(x).pred, y0)  # replay path
    sd2 = O.seeded_state_dict(model.state_dict(), m["seed"] + 1, cls_bias=m["cls_bias"])
    model.load_state_dict(sd2)
    y1 = pred.forward_device(x).pred.clone()
    fresh = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0)).forward_device(x).pred
    torch.cuda.synchronize()
    assert not torch.equal(y1, y0) and torch.equal(y1, fresh)
    with torch.no_grad():  # raw in-place edit of one parameter (what an optimizer kernel does, seen through torch)
        model.model[0].conv.weight.mul_(0.5)
    y2 = pred.forward_device(x).pred.clone()
    fresh2 = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0)).forward_device(x).pred
    assert torch.equal(y2, fresh2) and not torch.equal(y2, y1)
    model.train()
    model.eval()  # a training phase happened in between as far as the predictor can tell: packs dropped, plan re-recorded
    assert torch.equal(pred.forward_device(x).pred, fresh2)


def test_config4_tiled_scale_l_against_reference_rows(device):
    """BASELINE config 4 at its real size: Drone-YOLO-l, a 3840x2160 uint8 frame, eight 1280x1280 tiles (A = 136,000 per tile,
    ~13k candidates per tile through decode, filter and NMS), cross-tile merge.  Expectation: per-tile rows of the REAL
    reference + the oracle's merge (tests/golden/big.npz::l1280t8, computed once in the build container, 6.6 TFLOP of CPU work).
    fp32 storage: kept sets identical per tile and after the merge; fp16: the measured level."""
    import ast

    from drone_yolo_amd.engine.tiling import TiledPredictor, tile_offsets
    from drone_yolo_amd.utils import parity as PR

    g = golden("big.npz")
    meta, _, exp_rows, exp_idx = PR.golden_case("big.npz", "l1280t8")
    fr = ast.literal_eval(str(g["l1280t8__frame"]))
    hf, wf = fr["hw"]
    assert tile_offsets(hf, wf, fr["tile"], fr["overlap"]) == [tuple(o) for o in fr["offsets"]]
    frame = np.random.default_rng(fr["rng_seed"]).integers(0, 256, (hf, wf, 3), dtype=np.uint8)
    model = _bench_model(meta, device)
    exp_merged = g["l1280t8__merged"]
    for dtype in (torch.float32, H.F16X2, torch.float16):
        tp = TiledPredictor(model, tile=fr["tile"], overlap=fr["overlap"], merge_iou=fr["merge_iou"], merge_max_det=fr["merge_max_det"], conf=0.25, iou=0.7,
                            dtype=dtype, device=0)
        res = tp(frame)
        cf = tp.pred.forward_device(tp.last_tiles)  # the per-tile pass the merge consumed (replay of the recorded plan)
        torch.cuda.synchronize()
        par = PR.detection_parity(cf.nms, exp_rows, exp_idx)
        got = res.boxes.data.cpu().numpy()

        def iou_matrix(a, b):
            x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
            x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
            inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
            ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None] - inter
            return inter / np.maximum(ua, 1e-9)

        m = iou_matrix(exp_merged[:, :4].astype(np.float64), got[:, :4].astype(np.float64)) * (exp_merged[:, None, 5] == got[None, :, 5])
        merged_common = float((m.max(1) > 0.9).mean())  # merged reference boxes found again (same class, IoU > 0.9)
        _report("config4 l1280t8", {"dtype": str(dtype), **par, "merged": int(len(got)), "merged_ref": int(len(exp_merged)), "merged_common": merged_common})
        assert res.orig_shape == (hf, wf)
        if dtype in (torch.float32, H.F16X2):  # the bar: same kept sets per tile, IoU >= 0.999, and the same merged detections (fp32 and split float16)
            assert par["counts_equal"] and par["match_rate"] >= 0.999 and par["iou_min"] >= 0.999, par
            assert got.shape == exp_merged.shape and merged_common >= 0.995, merged_common
        else:  # fp16 storage: at most 1 % of the detections lost to near-tie flips
            assert par["missed_frac"] <= 0.01 and par["extra_frac"] <= 0.01 and par["iou_min"] >= 0.998, par
            assert merged_common >= 0.97, merged_common
        del tp, cf, res
        torch.cuda.empty_cache()


def test_config5_shape_scale_x_1536_against_reference_rows(device):
    """BASELINE config 5's model and shape (Drone-YOLO-x, 1536x1536, A = 195,840) in fp32 and fp16 storage against the rows the
    REAL reference computed in fp32 (tests/golden/big.npz::x1536): the expectation the fp8 path's tolerance is stated against."""
    from drone_yolo_amd.utils import parity as PR

    meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", "x1536")
    model = _bench_model(meta, device)
    for dtype in (torch.float32, H.F16X2, torch.float16):
        pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype=dtype, device=0))
        cf = pred.forward_device(pred.preprocess(x))
        torch.cuda.synchronize()
        par = PR.detection_parity(cf.nms, exp_rows, exp_idx)
        _report("config5-shape x1536", {"dtype": str(dtype), **par})
        if dtype in (torch.float32, H.F16X2):
            assert par["counts_equal"] and par["match_rate"] >= 0.999 and par["iou_min"] >= 0.999, par
        else:
            assert par["missed_frac"] <= 0.01 and par["extra_frac"] <= 0.01 and par["iou_min"] >= 0.998, par
        del pred, cf
        torch.cuda.empty_cache()


def _fp8_launches(cf):
    """(convolution launches on fp8 operands, all convolution launches) of a recorded pass."""
    from drone_yolo_amd import _lib

    convs = [args[0]._obj for fn, args, _ in cf.plan.ops if fn.__name__ == "dy_conv2d_nhwc"]
    return sum(1 for d in convs if d.dtype == _lib.DY_FP8), len(convs)


@pytest.mark.parametrize("tag", ["s640bench", "x1536"])
def test_config5_fp8_mixed_plan_meets_the_gate(tag, device):
    """BASELINE config 5 as a DEPLOYABLE precision ("match >= 0.90 and IoU min >= 0.98, not lowered to fit"):
    ``dtype="fp8-mixed"`` = float16 storage with the INTERNALS of every C2f block that does not feed the P2 Detect level in e4m3 on
    the block-scaled MFMA (BaseModel.fp8_plan_off_p2: layers 21, 24, 27 of the P2 YAMLs — cv1 16-bit -> fp8, Bottlenecks fp8, cv2
    fp8 -> 16-bit; every layer boundary and skip connection stays float16) — the set the error budget allows: profiles/r04_fp8_sensitivity_*.jsonl shows ONE e4m3 rounding on the P2
    path flipping 7-15 % of the kept boxes of these synthetic-weight fixtures, the whole deep neck < 1 %.  Against the rows the REAL
    reference computed in fp32 (tests/golden/big.npz): >= 90 % of the detections reproduced (same anchor AND class), <= 10 % extra,
    min IoU of the matched boxes >= 0.98; and the plan really runs fp8 (the blocks' Bottleneck and cv2 launches are on e4m3 operands)."""
    from drone_yolo_amd.utils import parity as PR

    meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", tag)
    model = _bench_model(meta, device)
    try:
        pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="fp8-mixed", device=0))
        cf = pred.forward_device(pred.preprocess(x))
        torch.cuda.synchronize()
        par = PR.detection_parity(cf.nms, exp_rows, exp_idx)
        n8, nconv = _fp8_launches(cf)
        _report(f"config5 fp8-mixed {tag}", {"dtype": "fp16 + e4m3 off the P2 path", "fp8_conv_launches": n8, "conv_launches": nconv, **par, **pred.fp8_calibration})
        assert bool(torch.isfinite(cf.pred).all())
        nb = sum(2 * len(model.model[i].m) + 1 for i in (21, 24, 27))
        assert pred.fp8_calibration["fp8_layers"] == [21, 24, 27] and n8 == nb, (pred.fp8_calibration, n8, nb)
        assert par["match_rate"] >= 0.90 and par["extra_frac"] <= 0.10 and par["iou_min"] >= 0.98, par
        cf2 = pred.forward_device(pred.preprocess(x))  # the recorded plan replays to the same result
        torch.cuda.synchronize()
        assert torch.equal(cf2.nms.count, cf.nms.count)
    finally:
        H.set_fp8_act_scale(1.0)


@pytest.mark.parametrize("tag", ["s640bench", "x1536"])
def test_config5_fp8_trunk_against_reference_rows(tag, device):
    """``dtype="fp8"``: the THROUGHPUT end of config 5 — every trunk layer stores e4m3 and runs on the block-scaled fp8 MFMA
    (csrc/conv_gemm_fk.hip; the 3-channel image layer in float16, the Detect branches float16 behind their first convolutions) —
    Drone-YOLO-x at 1536x1536 (A = 195,840) and Drone-YOLO-s at 640x640 against the rows the REAL reference computed in fp32.  This
    plan does NOT meet the deployable gate (the mixed plan above does) and the test says what it delivers instead, so that a kernel
    regression still shows: a 3-bit mantissa through ~60 layers of a synthetic-weight network whose kept scores sit within a few 0.01
    of each other reproduces >= 50 % of the reference detections (measured r02-r04: 0.57-0.59; the fake-quantised float16 pass of
    tools/fp8_sensitivity.py gives 0.589, i.e. the kernels add nothing to the format's own error), their boxes at mean IoU >= 0.975 and
    min IoU >= 0.90 (r04: 0.92-0.96 over 180-650 matched boxes), decoded boxes of all anchors within 1.5 px RMS, scores within 0.1."""
    from drone_yolo_amd.utils import parity as PR

    meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", tag)
    model = _bench_model(meta, device)
    try:
        pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="fp8", device=0))
        cf = pred.forward_device(pred.preprocess(x))
        torch.cuda.synchronize()
        par = PR.detection_parity(cf.nms, exp_rows, exp_idx)
        g = golden("big.npz")
        y_sub = cf.pred[:, :, ::199].cpu()
        ref_sub = torch.from_numpy(g[f"{tag}__y_sub"])
        box_rms = float(((y_sub[:, :4] - ref_sub[:, :4]) ** 2).mean().sqrt())
        cls_err = float((y_sub[:, 4:] - ref_sub[:, 4:]).abs().max())
        n8, nconv = _fp8_launches(cf)
        _report(f"config5 fp8 {tag}", {"dtype": "fp8_e4m3fn trunk, float16 Detect tail", "box_rms_px": box_rms, "cls_max_err": cls_err, "fp8_conv_launches": n8,
                                       "conv_launches": nconv, **par, **pred.fp8_calibration})
        assert bool(torch.isfinite(cf.pred).all()) and n8 >= 0.7 * nconv
        # min-IoU floor: 0.93 as set in round 3 on s640bench (measured 0.957 in r03 and r04).  x1536 is ONE image with 176 matched boxes: r03's
        # plan (everything e4m3, Detect too) had its worst box at 0.955, r04's plan (float16 Detect tails behind an e4m3 trunk) at 0.921 --
        # another plan, not the same arithmetic on a new kernel -- and is held to 0.90 (DESIGN §12 lists both measurements)
        iou_floor = 0.93 if tag == "s640bench" else 0.90
        assert par["match_rate"] >= 0.50 and par["iou_mean"] >= 0.975 and par["iou_min"] >= iou_floor and box_rms <= 1.5 and cls_err <= 0.1, (par, box_rms, cls_err)
    finally:
        H.set_fp8_act_scale(1.0)
