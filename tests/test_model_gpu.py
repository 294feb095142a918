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
    for dtype in (torch.float32, torch.float16):
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
        if dtype == torch.float32:  # the bar: same kept sets per tile, IoU >= 0.999, and the same merged detections
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
    for dtype in (torch.float32, torch.float16):
        pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype=dtype, device=0))
        cf = pred.forward_device(pred.preprocess(x))
        torch.cuda.synchronize()
        par = PR.detection_parity(cf.nms, exp_rows, exp_idx)
        _report("config5-shape x1536", {"dtype": str(dtype), **par})
        if dtype == torch.float32:
            assert par["counts_equal"] and par["match_rate"] >= 0.999 and par["iou_min"] >= 0.999, par
        else:
            assert par["missed_frac"] <= 0.01 and par["extra_frac"] <= 0.01 and par["iou_min"] >= 0.998, par
        del pred, cf
        torch.cuda.empty_cache()


@pytest.mark.parametrize("tag", ["s640bench", "x1536"])
def test_config5_fp8_storage_against_reference_rows(tag, device):
    """BASELINE config 5: fp8 (e4m3fn) weights and activations, fp8 MFMA with fp32 accumulate — Drone-YOLO-x at 1536x1536
    (A = 195,840) and, for scale, Drone-YOLO-s at 640x640 — against the rows the REAL reference computed in fp32
    (tests/golden/big.npz).  Bit-exact class / index parity is not expected under a 3-bit mantissa; the
    tolerance of this configuration, stated here and in DESIGN §2 (measured r02: match 0.58, IoU mean 0.982 - 0.987, min 0.955,
    box RMS 0.63 - 0.87 px, class-score error 0.04 - 0.06): at least 50 % of the reference detections reproduced (same anchor,
    same class — the synthetic network's scores sit within a few 0.01 of each other, so a 0.05 score error reorders many),
    their boxes at mean IoU >= 0.975 / min IoU >= 0.93, decoded boxes of all anchors within 1.5 px RMS, scores within 0.1."""
    from drone_yolo_amd.utils import parity as PR

    meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", tag)
    model = _bench_model(meta, device)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="fp8", device=0))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    par = PR.detection_parity(cf.nms, exp_rows, exp_idx)
    g = golden("big.npz")
    y_sub = cf.pred[:, :, ::199].cpu()
    ref_sub = torch.from_numpy(g[f"{tag}__y_sub"])
    box_rms = float(((y_sub[:, :4] - ref_sub[:, :4]) ** 2).mean().sqrt())
    cls_err = float((y_sub[:, 4:] - ref_sub[:, 4:]).abs().max())
    _report(f"config5 fp8 {tag}", {"dtype": "fp8_e4m3fn", "box_rms_px": box_rms, "cls_max_err": cls_err, **par, **pred.fp8_calibration})
    assert bool(torch.isfinite(cf.pred).all())
    assert par["match_rate"] >= 0.50 and par["iou_mean"] >= 0.975 and par["iou_min"] >= 0.93 and box_rms <= 1.5 and cls_err <= 0.1, (par, box_rms, cls_err)
    H.set_fp8_act_scale(1.0)


def test_replay_graph_and_api(device):
    """LaunchPlan replay and hipGraph replay reproduce the recorded pass bit for bit; YOLO.predict API shape."""
    g = golden("e2e.npz")
    m, d, sd, model, x = _build("n128", g, device)
    outs = []
    for graph in (False, True):
        pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0, graph=graph))
        xin = pred.preprocess(x)
        cf = pred.forward_device(xin)
        torch.cuda.synchronize()
        first = (cf.pred.clone(), cf.nms.out.clone(), cf.nms.count.clone())
        cf.pred.zero_(), cf.nms.out.zero_(), cf.nms.count.zero_()
        cf2 = pred.forward_device(xin.clone())  # different input buffer, same contents
        torch.cuda.synchronize()
        assert cf2 is cf
        assert torch.equal(cf.pred, first[0]) and torch.equal(cf.nms.out, first[1]) and torch.equal(cf.nms.count, first[2])
        outs.append(first)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    yolo = D.YOLO("yolov8n-p2-repvgg.yaml")
    yolo.model.load_state_dict(sd, strict=False) if yolo.model.yaml["nc"] == m["nc"] else None
    res = yolo.predict(torch.rand(2, 3, 64, 96), device=0, dtype="fp32", conf=0.001)
    assert len(res) == 2 and res[0].boxes.data.shape[1] == 6 and res[0].orig_shape == (64, 96)
    assert res[0].boxes.xyxy.shape[1] == 4 and res[0].boxes.xywhn.shape == res[0].boxes.xyxy.shape
    assert set(res[0].speed) == {"preprocess", "inference", "postprocess"}
    with pytest.raises(RuntimeError):
        yolo.predict(torch.rand(1, 3, 64, 64), device="cpu")


def test_full_size_properties(device):
    """Drone-YOLO-s 640x640 at the bench batch: size-independent properties (no oracle at this size).
    (1) images are independent: a batch equals its images run one by one; (2) permuting the batch
    permutes the outputs; (3) NMS output invariants: counts <= max_det, scores sorted descending,
    boxes inside the image, kept anchors unique."""
    g = golden("e2e.npz")
    m, d, sd, model, _ = _build("s640", g, device)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0))
    x = torch.rand(8, 3, 640, 640, generator=torch.Generator().manual_seed(5)).to(device)
    cf = pred.forward_device(x)
    torch.cuda.synchronize()
    out, cnt, idx, y = cf.nms.out.clone(), cf.nms.count.clone(), cf.nms.index.clone(), cf.pred.clone()
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=device)
    cf = pred.forward_device(x[perm].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(cf.pred, y[perm]) and torch.equal(cf.nms.out, out[perm]) and torch.equal(cf.nms.count, cnt[perm])
    single = pred.forward_device(x[2:3].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(single.pred[0], y[2]) and torch.equal(single.nms.out[0], out[2])
    for i in range(8):
        c = int(cnt[i])
        assert 0 < c <= 300
        sc = out[i, :c, 4]
        assert bool((sc[:-1] >= sc[1:]).all()) and float(sc.min()) > 0.25
        assert float(out[i, :c, :4].min()) >= 0 and float(out[i, :c, :4].max()) <= 640
        assert len(set(idx[i, :c].tolist())) == c


def test_image_sources_letterbox_to_results(device):
    """Array sources end to end: list of BGR uint8 frames -> LetterBox kernel -> model -> NMS -> boxes mapped back to the
    original frame (scale_boxes), against the oracle chain (letterbox_oracle.preprocess -> forward -> NMS -> scale_boxes)."""
    from oracle import letterbox_oracle as LB

    g = golden("e2e.npz")
    m, d, sd, model, _ = _build("n128", g, device)
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (180, 300, 3), dtype=np.uint8) for _ in range(2)]
    x = torch.from_numpy(LB.preprocess(frames, (128, 128), auto=True, stride=32))
    assert tuple(x.shape) == (2, 3, 96, 128)  # minimum rectangle: 300x180 -> 128x77 + 19 rows of padding (mod 32)
    with torch.no_grad():
        y, _ = O.forward(d, sd, x)
    det, _ = O.non_max_suppression(y, 0.25, 0.7, max_det=300, nc=m["nc"], return_index=True)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.float32, device=0, imgsz=128))
    res = pred(frames)
    assert len(res) == 2 and res[0].orig_shape == (180, 300)
    for i in range(2):
        exp = det[i].clone()
        exp[:, :4] = O.scale_boxes(x.shape[2:], exp[:, :4], (180, 300))
        got = res[i].boxes.data.cpu()
        assert got.shape == exp.shape and len(exp) > 0
        assert torch.equal(got[:, 5], exp[:, 5]) and torch.allclose(got[:, :5], exp[:, :5], atol=2e-2, rtol=1e-4)
    # a second call with frames of another shape re-uses the recorded pass only when the letterboxed size matches
    frames2 = [rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)]
    res2 = pred(frames2)
    assert res2[0].orig_shape == (128, 128)


@pytest.mark.parametrize("scale", ["m", "l", "x"])
def test_other_scales_match_oracle(scale, device):
    """The wider/deeper scales of the YAML (m, l, x: channel widths 48..640, repeats up to 3; SURVEY §8d configs 4-5 use l
    and x) through the same kernels, fp32 storage, against the oracle on seeded weights."""
    d = load_yaml("yolov8-p2-repvgg.yaml", scale, 10)
    model = D.DetectionModel(dict(d), nc=10, verbose=False)
    sd = O.seeded_state_dict(model.state_dict(), 300 + ord(scale), cls_bias=-2.0)
    model.load_state_dict(sd)
    x = torch.rand(2, 3, 96, 64, generator=torch.Generator().manual_seed(ord(scale)))
    with torch.no_grad():
        y, _ = O.forward(d, sd, x)
    det, idx = O.non_max_suppression(y, 0.25, 0.7, max_det=300, nc=10, return_index=True)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.float32, device=0))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    err = float((cf.pred.cpu() - y).abs().max())
    assert err < 5e-2, err
    counts = cf.nms.count.cpu().tolist()
    assert counts == [len(r) for r in det] and sum(counts) > 0
    for i, c in enumerate(counts):
        assert np.array_equal(np.sort(cf.nms.index[i, :c].cpu().numpy()), np.sort(idx[i].numpy()))
    # and the throughput dtype runs
    pred16 = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0))
    cf16 = pred16.forward_device(pred16.preprocess(x))
    torch.cuda.synchronize()
    e16 = (cf16.pred.cpu() - y).abs()
    # untrained seeded weights: a few P5 anchors (stride 32, boxes ~100 px wide) move by several pixels in bf16
    assert bool(torch.isfinite(cf16.pred).all()) and float(e16.median()) < 0.05 and float(e16[:, 4:].max()) < 0.4, (float(e16.median()), float(e16.max()))


def test_yolo_from_reference_checkpoint(device):
    """YOLO('<reference-pickled>.pt').predict on the device reproduces what the reference computed from that checkpoint."""
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, "tests", "golden", "ref_checkpoint_t.npz"))
    yolo = D.YOLO(os.path.join(root, "tests", "golden", "ref_checkpoint_t.pt"))
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    pred = D.engine.predictor.DetectionPredictor(yolo.model, dict(conf=0.25, iou=0.7, dtype=torch.float32, device=0))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    assert torch.allclose(cf.pred.cpu(), torch.from_numpy(g["y"]), atol=5e-3, rtol=1e-4)


def test_tiled_inference_matches_oracle_chain(device):
    """Tile slicer + per-tile pass + cross-tile merge NMS on the device against the same chain on the CPU oracle
    (numpy crops -> O.forward -> O.non_max_suppression per tile -> shift -> O.non_max_suppression over the union)."""
    from drone_yolo_amd.engine.tiling import TiledPredictor, tile_offsets

    assert tile_offsets(2160, 3840, 1280, 0.2) == [(y, x) for y in (0, 880) for x in (0, 1024, 2048, 2560)]
    assert tile_offsets(100, 100, 128, 0.2) == [(0, 0)]
    g = golden("e2e.npz")
    m, d, sd, model, _ = _build("n128", g, device)
    rng = np.random.default_rng(21)
    frame = rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)
    tile, nc = 128, m["nc"]
    offs = tile_offsets(200, 300, tile, 0.25)
    assert len(offs) == 2 * 3
    x = torch.stack([torch.from_numpy(np.ascontiguousarray(frame[y : y + tile, xx : xx + tile, ::-1].transpose(2, 0, 1))).float() / 255 for y, xx in offs])
    with torch.no_grad():
        yy, _ = O.forward(d, sd, x)
    det, _ = O.non_max_suppression(yy, 0.25, 0.7, max_det=300, nc=nc, return_index=True)
    rows = []
    for (oy, ox), r in zip(offs, det):
        r = r.clone()
        r[:, :4] = O.clip_boxes(r[:, :4], (tile, tile))  # each tile is an image of its own to the predictor (detect/predict.py:59-73)
        r[:, [0, 2]] += ox
        r[:, [1, 3]] += oy
        rows.append(r)
    allr = torch.cat(rows)
    pred = torch.zeros(1, 4 + nc, len(allr))
    pred[0, 0], pred[0, 1] = (allr[:, 0] + allr[:, 2]) / 2, (allr[:, 1] + allr[:, 3]) / 2
    pred[0, 2], pred[0, 3] = allr[:, 2] - allr[:, 0], allr[:, 3] - allr[:, 1]
    pred[0, 4 + allr[:, 5].long(), torch.arange(len(allr))] = allr[:, 4]
    merged, _ = O.non_max_suppression(pred, 0.0, 0.6, max_det=1000, nc=nc, return_index=True)
    exp = merged[0]
    tp = TiledPredictor(model, tile=tile, overlap=0.25, merge_iou=0.6, conf=0.25, iou=0.7, dtype=torch.float32, device=0)
    res = tp(frame)
    got = res.boxes.data.cpu()
    assert res.orig_shape == (200, 300) and 0 < len(exp) < len(allr)  # the merge removed cross-tile duplicates
    assert got.shape == exp.shape, (got.shape, exp.shape)
    assert torch.equal(got[:, 5], exp[:, 5]) and torch.allclose(got[:, :5], exp[:, :5], atol=3e-2, rtol=1e-4)


def test_two_batches_in_flight_on_two_streams(device):
    """bench.py keeps two batches in flight on two HIP streams (separate predictor state and hipGraph each): the results
    must be exactly what the same batches give one after the other on one stream."""
    g = golden("e2e.npz")
    m, d, sd, model, _ = _build("n128", g, device)
    xs = [torch.rand(4, 3, 128, 96, generator=torch.Generator().manual_seed(40 + j)).to(device) for j in range(2)]
    serial = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0, graph=True))
    ref = []
    for x in xs:
        cf = serial.forward_device(x)
        torch.cuda.synchronize()
        ref.append((cf.pred.clone(), cf.nms.out.clone(), cf.nms.count.clone()))
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]
    preds, cfs = [], []
    for j in range(2):
        with torch.cuda.stream(streams[j]):
            pj = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0, graph=True))
            cfs.append(pj.forward_device(xs[j]))
            preds.append(pj)
    torch.cuda.synchronize()
    for _ in range(6):  # interleaved replays
        for j in range(2):
            with torch.cuda.stream(streams[j]):
                preds[j].forward_device(xs[j])
    torch.cuda.synchronize()
    for j in range(2):
        assert torch.equal(cfs[j].pred, ref[j][0]) and torch.equal(cfs[j].nms.out, ref[j][1]) and torch.equal(cfs[j].nms.count, ref[j][2])


def test_fusions_agree_with_layer_by_layer_path(device):
    """Drone-YOLO-s 640x640 bf16: the one-launch forms (layers 0 + 1 fused, stride-4 C2f fused, Detect first convs stacked)
    against the same model run layer by layer.  Same operands and rounding points, different K summation order: raw
    predictions agree to a few bf16 roundings of the deepest activations and the kept detections match."""
    g = golden("e2e.npz")
    m, d, sd, model, _ = _build("s640", g, device)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0))
    x = torch.rand(2, 3, 640, 640, generator=torch.Generator().manual_seed(11)).to(device)
    det = model.model[-1]
    cf = pred.forward_device(x)
    torch.cuda.synchronize()
    y1, o1, c1 = cf.pred.clone(), cf.nms.out.clone(), cf.nms.count.clone()
    model.fuse_stem2 = False
    det.fuse_first = False
    for mod in model.modules():
        if isinstance(mod, M.C2f):
            mod.fuse_block = False
    pred2 = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.bfloat16, device=0))
    cf2 = pred2.forward_device(x)
    torch.cuda.synchronize()
    y2, o2, c2 = cf2.pred, cf2.nms.out, cf2.nms.count
    box_err = float((y1[:, :4] - y2[:, :4]).abs().max())
    cls_err = float((y1[:, 4:] - y2[:, 4:]).abs().max())
    matched = 0
    total = 0
    for i in range(2):
        a, b = o1[i, : int(c1[i])], o2[i, : int(c2[i])]
        total += max(len(a), len(b))
        if len(a) and len(b):
            an, bn = a.float().cpu().numpy(), b.float().cpu().numpy()
            iou = box_iou_pairs(np.repeat(an[:, :4], len(bn), 0), np.tile(bn[:, :4], (len(an), 1))).reshape(len(an), len(bn))
            same = an[:, None, 5] == bn[None, :, 5]
            matched += int(((iou > 0.95) & same).any(1).sum())
    _report("fused_vs_layerwise_s640", {"box_max_abs_px": box_err, "cls_max_abs": cls_err, "matched": matched, "total": total})
    assert box_err < 8.0 and cls_err < 0.08, (box_err, cls_err)
    assert total > 0 and matched >= 0.9 * total, (matched, total)
    model.fuse_stem2 = True
    det.fuse_first = True
