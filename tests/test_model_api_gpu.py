"""GPU tests of the recorded pass and the public API around it: LaunchPlan / hipGraph replay, batch independence at full size, array
sources through the LetterBox kernel, the m / l / x scales against the oracle, a reference-pickled checkpoint, the tiled chain against
the oracle chain, two batches in flight on two streams, and the fused launches against the layer-by-layer path.

These eight tests were dropped from tests/test_model_gpu.py in round 4 (commit 707f7a7) without a replacement; they are restored here
unchanged."""
import numpy as np
import pytest
import torch

import drone_yolo_amd as D
from drone_yolo_amd.nn import modules as M
from oracle import drone_yolo_oracle as O
from tests._util import box_iou_pairs, golden, load_yaml, split_rows
from tests.test_model_gpu import _build, _report

pytestmark = pytest.mark.gpu


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
        # r05: another batch at another address (graph: the image launch runs from the caller's address, the rest from the tail graph — no copy
        # into the static input), then the first batch again from ITS address, then from the graph's own static input
        x2 = torch.rand(xin.shape, generator=torch.Generator().manual_seed(11)).to(device)
        pred.forward_device(x2)
        torch.cuda.synchronize()
        other = (cf.pred.clone(), cf.nms.out.clone(), cf.nms.count.clone())
        assert not torch.equal(other[0], first[0])
        for again in (xin, cf.static_in if graph else xin):
            if graph and again is cf.static_in:
                cf.static_in.copy_(xin)
            cf.pred.zero_(), cf.nms.out.zero_(), cf.nms.count.zero_()
            pred.forward_device(again)
            torch.cuda.synchronize()
            assert torch.equal(cf.pred, first[0]) and torch.equal(cf.nms.out, first[1]) and torch.equal(cf.nms.count, first[2])
        if graph:
            assert cf.graph_tail is not None
        outs.append(first + other)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][3], outs[1][3]) and torch.equal(outs[0][4], outs[1][4]) and torch.equal(outs[0][5], outs[1][5])
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


def test_yolo_predict_with_default_arguments_reproduces_the_reference_rows(device):
    """``YOLO(yaml).predict(x)`` with NO precision argument — what a user of the drop-in gets — against the rows the
    REAL reference computed on the CPU in fp32 for BASELINE config 2's fixture (tests/golden/big.npz::s640bench, 1,103 detections): same
    kept anchors, same classes, IoU >= 0.999.  The reference's default is ``half: False`` (cfg/default.yaml:54); here that selects the
    bar-exact precision and a hipGraph replay (engine/predictor.py::EXACT_DTYPE)."""
    from drone_yolo_amd.engine.predictor import EXACT_DTYPE
    from drone_yolo_amd.utils import parity as PR
    from tests.test_model_gpu import _bench_model

    meta, x, exp_rows, exp_idx = PR.golden_case("big.npz", "s640bench")
    yolo = D.YOLO("yolov8s-p2-repvgg.yaml")
    yolo.model = _bench_model(meta, device)  # the fixture's weights (bench.py's recipe, nc = 10)
    res = yolo.predict(x)
    assert yolo.predictor.dtype == EXACT_DTYPE and yolo.predictor.args["graph"] and yolo.predictor.args["half"] is False
    cf = next(iter(yolo.predictor._compiled.values()))
    assert cf.graph is not None
    par = PR.detection_parity(cf.nms, exp_rows, exp_idx, conf=0.25, margin=0.0)
    assert par["counts_equal"] and par["kept_sets_identical"] and par["match_rate"] == 1.0 and par["iou_min"] >= 0.999, par
    assert [len(r) for r in res] == [len(r) for r in exp_rows]
    for i, (r, e) in enumerate(zip(res, exp_rows)):  # the Results the API hands back carry the same rows (aligned by kept anchor: near-tied scores may swap ranks)
        got = r.boxes.data.cpu().numpy()
        at = {int(a): k for k, a in enumerate(cf.nms.index[i, : len(got)].cpu().tolist())}
        got = got[[at[int(a)] for a in exp_idx[i]]]
        assert np.array_equal(got[:, 5], e[:, 5]) and np.allclose(got[:, :5], e[:, :5], atol=2e-2)
    res2 = yolo.predict(x)  # the second call replays the graph: identical rows
    assert all(torch.equal(a.boxes.data, b.boxes.data) for a, b in zip(res, res2))
    half = yolo.predict(x, half=True)
    assert yolo.predictor.dtype == torch.float16 and len(half) == len(res)


def test_image_sources_of_different_shapes_are_letterboxed_per_image(device):
    """VERDICT r4 missing 4 (reference engine/predictor.py:147-163): sources of different shapes -> every image letterboxed on its own to
    imgsz x imgsz (auto = False), one batch, boxes mapped back per image — against the oracle chain run image by image."""
    from oracle import letterbox_oracle as LB

    g = golden("e2e.npz")
    m, d, sd, model, _ = _build("n128", g, device)
    rng = np.random.default_rng(17)
    frames = [rng.integers(0, 256, (180, 300, 3), dtype=np.uint8), rng.integers(0, 256, (250, 140, 3), dtype=np.uint8), rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)]
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, dtype=torch.float32, device=0, imgsz=128))
    res = pred(frames)
    assert [r.orig_shape for r in res] == [(180, 300), (250, 140), (128, 128)]
    for i, f in enumerate(frames):
        x = torch.from_numpy(LB.preprocess([f], (128, 128), auto=False, stride=32))
        assert tuple(x.shape) == (1, 3, 128, 128)
        with torch.no_grad():
            y, _ = O.forward(d, sd, x)
        det, _ = O.non_max_suppression(y, 0.25, 0.7, max_det=300, nc=m["nc"], return_index=True)
        exp = det[0].clone()
        exp[:, :4] = O.scale_boxes(x.shape[2:], exp[:, :4], f.shape[:2])
        got = res[i].boxes.data.cpu()
        assert got.shape == exp.shape and len(exp) > 0, (i, got.shape, exp.shape)
        assert torch.equal(got[:, 5], exp[:, 5]) and torch.allclose(got[:, :5], exp[:, :5], atol=2e-2, rtol=1e-4), i


def test_yolo_profile_reports_every_launching_layer(device):
    """``YOLO.profile(x)`` (reference predict(profile=True) -> _profile_one_layer, nn/tasks.py:171-191): per-layer device time of the recorded pass; the rows
    cover every launch of the plan, name the layers that launch (folded Upsample / Concat do not) and end with the postprocess row."""
    yolo = D.YOLO("yolov8n-p2-repvgg.yaml")
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(2))
    rows = yolo.profile(x, device=0, conf=0.001)
    cf = next(iter(yolo.predictor._compiled.values()))
    assert sum(r["launches"] for r in rows) == len(cf.plan.ops) and all(r["ms"] > 0 for r in rows)
    types = [r["type"] for r in rows]
    assert types[-1].startswith("postprocess") and "Detect" in types and "SPPF" in types and "C2f" in types and "Upsample" not in types
    assert [r["layer"] for r in rows[:-1]] == sorted(r["layer"] for r in rows[:-1])


def test_default_precision_follows_what_the_split_kernels_cover(device):
    """The bar-exact default is split float16 where its kernels cover the model — r05: the -sf YAML too, its DWConv on the split small-group kernel —
    and fp32 storage otherwise (here: a model given a 5x5 convolution); half=True is float16."""
    from drone_yolo_amd import hip_ops as H
    from drone_yolo_amd.engine.predictor import split_supported

    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    for yaml_name, want in (("yolov8n-p2-repvgg.yaml", H.F16X2), ("yolov8n-p2-repvgg-sf.yaml", H.F16X2), ("yolov8n.yaml", H.F16X2)):
        yolo = D.YOLO(yaml_name)
        assert split_supported(yolo.model) == (want == H.F16X2)
        res = yolo.predict(x, device=0, conf=0.001)
        assert yolo.predictor.dtype == want and len(res) == 1, (yaml_name, yolo.predictor.dtype)
        ref = yolo.predict(x, device=0, conf=0.001, dtype="fp32")  # the two bar-exact precisions agree on the rows
        assert res[0].boxes.data.shape == ref[0].boxes.data.shape
        assert torch.allclose(res[0].boxes.data[:, :4], ref[0].boxes.data[:, :4], atol=2e-2) and torch.equal(res[0].boxes.data[:, 5], ref[0].boxes.data[:, 5])
    odd = D.YOLO("yolov8n-p2-repvgg.yaml")
    conv = odd.model.model[0].conv
    odd.model.model[0].conv = torch.nn.Conv2d(conv.in_channels, conv.out_channels, 5, 2, 2, bias=False)  # not a kernel size the split kernels take
    assert not split_supported(odd.model)
    assert D.YOLO("yolov8n-p2-repvgg.yaml").predict(x, device=0, half=True) is not None


def test_a_source_larger_than_one_batch_is_streamed_batch_by_batch(device):
    """`predict(source, batch=B)` with more images than B (reference stream_inference, predictor.py:221-298: `for self.batch in self.dataset`): the
    images run B at a time — a ragged last batch included — one `Results` per image in order, `stream=True` as a generator; the device works a batch
    ahead of the host, and the rows are those of the same images run as ONE batch (batch independence), for a float tensor and for uint8 frames."""
    import types

    g = golden("e2e.npz")
    m, d, sd, model, x = _build("n128", g, device)
    yolo = D.YOLO("yolov8n-p2-repvgg.yaml")
    yolo.model = model
    gen = torch.Generator().manual_seed(21)
    xs = torch.rand(10, 3, 128, 128, generator=gen)
    whole = yolo.predict(xs, device=0, conf=0.05)
    assert len(whole) == 10
    for stream in (False, True):
        got = yolo.predict(xs, device=0, conf=0.05, batch=4, stream=stream)
        assert isinstance(got, types.GeneratorType) == stream
        got = list(got)
        assert len(got) == 10 and all(set(r.speed) == {"preprocess", "inference", "postprocess"} for r in got)
        for a, b in zip(got, whole):
            assert torch.equal(a.boxes.data, b.boxes.data) and a.orig_shape == b.orig_shape
        assert [r.path for r in got] == [f"image{i}.jpg" for i in range(10)]
    frames = [np.ascontiguousarray((torch.rand(96, 120, 3, generator=gen) * 255).to(torch.uint8).numpy()) for _ in range(7)]
    whole = yolo.predict(frames, device=0, conf=0.05)
    got = list(yolo.predict(frames, device=0, conf=0.05, batch=3, stream=True))
    assert len(got) == 7
    for a, b in zip(got, whole):
        assert torch.equal(a.boxes.data, b.boxes.data) and a.orig_shape == (96, 120)


def test_image_files_and_pil_images_as_sources(device, tmp_path):
    """`predict("dir")` / `predict(["a.png", ...])` / `predict(PIL image)` (reference check_source -> LoadImagesAndVideos / LoadPilAndNumpy): files are
    decoded on the host and run `batch` at a time — by default one by one, each letterboxed to its own minimum rectangle, as the reference's loader
    does — and give what the decoded frames give; `Results.path` / `orig_img` / `orig_shape` are the file's."""
    from PIL import Image

    g = golden("e2e.npz")
    m, d, sd, model, x = _build("n128", g, device)
    yolo = D.YOLO("yolov8n-p2-repvgg.yaml")
    yolo.model = model
    gen = torch.Generator().manual_seed(31)
    frames = []
    for i, (h, w) in enumerate(((96, 120), (128, 128), (70, 100), (96, 120))):
        f = np.ascontiguousarray((torch.rand(h, w, 3, generator=gen) * 255).to(torch.uint8).numpy())
        frames.append(f)
        Image.fromarray(f[:, :, ::-1].copy()).save(tmp_path / f"img{i}.png")  # (files hold RGB; frames are BGR)
    res = yolo.predict(str(tmp_path), device=0, conf=0.05)
    assert [r.path.rsplit("/", 1)[-1] for r in res] == [f"img{i}.png" for i in range(4)]
    for i, r in enumerate(res):
        one = yolo.predict([frames[i]], device=0, conf=0.05)[0]  # the same image as an array source: a batch of its own
        assert torch.equal(r.boxes.data, one.boxes.data) and r.orig_shape == frames[i].shape[:2] and np.array_equal(r.orig_img, frames[i])
    # `batch=3`: three files at a time (different shapes in one batch: each letterboxed to the full size), then the ragged rest
    got = list(yolo.predict([str(tmp_path / f"img{i}.png") for i in range(4)], device=0, conf=0.05, batch=3, stream=True))
    ref = yolo.predict(frames[:3], device=0, conf=0.05) + yolo.predict(frames[3:], device=0, conf=0.05)
    assert len(got) == 4 and all(torch.equal(a.boxes.data, b.boxes.data) for a, b in zip(got, ref))
    # PIL images (LoadPilAndNumpy): one batch of all of them, names from `filename`
    with Image.open(tmp_path / "img0.png") as a, Image.open(tmp_path / "img3.png") as b:
        pres = yolo.predict([a, b], device=0, conf=0.05)
        assert pres[0].path.endswith("img0.png")
    ref = yolo.predict([frames[0], frames[3]], device=0, conf=0.05)
    assert all(torch.equal(a.boxes.data, b.boxes.data) for a, b in zip(pres, ref))


@pytest.mark.parametrize("tag", ["n128", "n64"])
def test_augmented_inference_matches_the_reference_rows(tag, device):
    """`predict(x, augment=True)` (reference DetectionModel._predict_augment, nn/tasks.py:347-383) on the device in fp32 storage: the image pyramid from
    `dy_scale_img_nchw_f32`, three passes of the path, de-scaled / de-mirrored / clipped / merged — the merged output to fp32 round-off of the REAL
    reference's (tests/golden/aug.npz) and the rows its NMS keeps: the same anchors, classes and order."""
    g = golden("aug.npz")
    m, d, sd, model, x = _build(tag, golden("e2e.npz"), device)
    pred = D.engine.predictor.DetectionPredictor(model, dict(conf=0.25, iou=0.7, max_det=300, dtype="fp32", device=0, augment=True))
    cf = pred.forward_device(pred.preprocess(x))
    torch.cuda.synchronize()
    yref = torch.from_numpy(g[f"{tag}__y"])
    y = cf.pred.cpu()
    assert tuple(y.shape) == tuple(yref.shape)
    assert float((y[:, :4] - yref[:, :4]).abs().max()) < 2e-2 and float((y[:, 4:] - yref[:, 4:]).abs().max()) < 1e-4
    counts = cf.nms.count.cpu().tolist()
    assert counts == [int(v) for v in g[f"{tag}__n"]]
    exp_idx = split_rows(g[f"{tag}__det_idx"], g[f"{tag}__n"])
    exp_rows = split_rows(g[f"{tag}__det"], g[f"{tag}__n"])
    for i, c in enumerate(counts):
        got_idx = cf.nms.index[i, :c].cpu().numpy()
        assert sorted(got_idx.tolist()) == sorted(exp_idx[i].tolist())
        score_of = {int(a): float(s) for a, s in zip(exp_idx[i], exp_rows[i][:, 4])}
        for k in np.nonzero(got_idx != exp_idx[i])[0]:  # two detections whose scores differ by fp32 round-off may swap places
            assert abs(score_of[int(got_idx[k])] - float(exp_rows[i][k, 4])) < 2e-6
    # the public call, at the default precision as well
    yolo = D.YOLO("yolov8n-p2-repvgg.yaml")
    yolo.model = model
    res = yolo.predict(x, device=0, augment=True)
    assert [len(r) for r in res] == counts
    with pytest.raises(NotImplementedError):
        yolo.predict(x, device=0, augment=True, dtype="fp8")
