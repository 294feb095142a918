"""CPU: the oracle restatement reproduces the golden vectors captured from the real reference
(oracle/make_golden.py).  This is what pins the oracle on machines without /root/reference."""
import ast

import numpy as np
import pytest
import torch

from oracle import drone_yolo_oracle as O
from tests._util import golden, load_yaml, meta, split_rows


def _sd_for(module_state_template, seed, prefix="model.0."):
    sd = O.seeded_state_dict(module_state_template, seed)
    return {prefix + k: v for k, v in sd.items()}


def _template(mod):
    return {k: v for k, v in mod.state_dict().items()}


def test_per_op_vectors():
    import drone_yolo_amd.nn.modules as M  # only used as a shape/key template for the seeded weights

    g = golden("per_op.npz")
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    c1, c2, k, s = g["conv_args"]
    sd = _sd_for(_template(M.Conv(int(c1), int(c2), int(k), int(s))), int(g["conv_seed"]))
    assert torch.allclose(O.conv_block(t("conv_x"), sd, "model.0", int(k), int(s)), t("conv_y"), atol=1e-4)
    c1, c2, k, s = g["conv1_args"]
    sd = _sd_for(_template(M.Conv(int(c1), int(c2), int(k), int(s))), int(g["conv1_seed"]))
    assert torch.allclose(O.conv_block(t("conv1_x"), sd, "model.0", 1, 1), t("conv1_y"), atol=1e-4)
    c1, c2, k, s = g["dw_args"]
    sd = _sd_for(_template(M.DWConv(int(c1), int(c2), int(k), int(s))), int(g["dw_seed"]))
    assert torch.allclose(O.conv_block(t("dw_x"), sd, "model.0", 3, 2, g=16), t("dw_y"), atol=1e-4)
    for tag in ("rep_s2", "rep_id"):
        c1, c2, k, s = (int(v) for v in g[f"{tag}_args"])
        sd = _sd_for(_template(M.RepVGGBlock(c1, c2, 3, s)), int(g[f"{tag}_seed"]))
        y = O.repvgg_block(t(f"{tag}_x"), sd, "model.0", s, has_identity=(s == 1))
        assert torch.allclose(y, t(f"{tag}_y"), atol=1e-4)
        kk, bb = O.repvgg_equivalent(sd, "model.0", s == 1, c1)
        yf = torch.nn.functional.silu(torch.nn.functional.conv2d(t(f"{tag}_x"), kk, bb, s, 1))
        assert torch.allclose(yf, t(f"{tag}_y"), atol=2e-4)
    sd = _sd_for(_template(M.Bottleneck(16, 16, True, 1, k=((3, 3), (3, 3)), e=1.0)), int(g["bott_seed"]))
    assert torch.allclose(O.bottleneck(t("bott_x"), sd, "model.0", True, True), t("bott_y"), atol=1e-4)
    for tag in ("c2f_a", "c2f_b"):
        c1, c2, n, sc = (int(v) for v in g[f"{tag}_args"])
        sd = _sd_for(_template(M.C2f(c1, c2, n, bool(sc))), int(g[f"{tag}_seed"]))
        assert torch.allclose(O.c2f(t(f"{tag}_x"), sd, "model.0", n, bool(sc), True), t(f"{tag}_y"), atol=3e-4)
    sd = _sd_for(_template(M.SPPF(32, 32, 5)), int(g["sppf_seed"]))
    assert torch.allclose(O.sppf(t("sppf_x"), sd, "model.0", 5, True), t("sppf_y"), atol=3e-4)
    assert torch.allclose(O.dfl(t("dfl_x")), t("dfl_y"), atol=1e-5)
    pts, st = O.make_anchors([(6, 8), (3, 4)], [8, 16])
    assert torch.equal(pts, t("anchors_pts")) and torch.equal(st, t("anchors_st"))
    assert torch.allclose(O.dist2bbox(t("d2b_x"), pts.t().unsqueeze(0), True, 1), t("d2b_y"), atol=1e-6)
    assert torch.equal(O.xywh2xyxy(t("xywh_x")), t("xywh_y"))
    h1, w1, h0, w0 = (int(v) for v in g["scale_shapes"])
    assert torch.allclose(O.scale_boxes((h1, w1), t("scale_x").clone(), (h0, w0)), t("scale_y"), atol=1e-5)
    M.Detect.legacy = True
    det = M.Detect(nc=5, ch=(16, 32))
    sd = _sd_for(_template(det), int(g["det_seed"]))
    raw = O.detect_head([t("det_x0"), t("det_x1")], sd, "model.0", 5, True)
    assert torch.allclose(raw[0], t("det_raw0"), atol=1e-4) and torch.allclose(raw[1], t("det_raw1"), atol=1e-4)
    assert torch.allclose(O.detect_decode(raw, [8.0, 16.0], 5), t("det_y"), atol=2e-3)


def test_nms_vectors():
    g = golden("nms.npz")
    names = sorted({k.split("__")[0] for k in g.files})
    assert len(names) >= 8
    for name in names:
        pred = torch.from_numpy(g[f"{name}__pred"])
        kw = ast.literal_eval(str(g[f"{name}__kw"]))
        out = O.non_max_suppression(pred, **kw)
        exp = split_rows(g[f"{name}__out"], g[f"{name}__n"])
        assert [len(o) for o in out] == [len(e) for e in exp], name
        for o, e in zip(out, exp):
            assert np.array_equal(o.numpy(), e), name


def test_nms_multilabel_vectors():
    """The validator's NMS (multi_label=True, ops.py:286-288) restated by the oracle against the REAL reference's rows (nms_ml.npz)."""
    g = golden("nms_ml.npz")
    names = sorted({k.split("__")[0] for k in g.files})
    assert len(names) >= 8
    for name in names:
        pred = torch.from_numpy(g[f"{name}__pred"])
        kw = ast.literal_eval(str(g[f"{name}__kw"]))
        out, idx = O.non_max_suppression(pred, return_index=True, **kw)
        exp = split_rows(g[f"{name}__out"], g[f"{name}__n"])
        assert [len(o) for o in out] == [len(e) for e in exp], name
        for o, e in zip(out, exp):
            assert np.array_equal(o.numpy(), e), name
        assert np.array_equal(np.concatenate([t.numpy() for t in idx]) if len(idx) else np.zeros(0), g[f"{name}__idx"]), name


def test_nms_greedy_properties():
    """Brute-force properties of greedy NMS (the boundary the reference's own tests do not pin):
    kept boxes are pairwise IoU <= thr within a class; every dropped box overlaps an earlier kept one."""
    rng = np.random.default_rng(3)
    for trial in range(20):
        n = int(rng.integers(1, 120))
        xy = rng.random((n, 2), dtype=np.float32) * 60
        wh = rng.random((n, 2), dtype=np.float32) * 30 + 1
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        scores = np.round(rng.random(n, dtype=np.float32), 2)  # rounding makes ties common
        thr = float(rng.choice([0.3, 0.5, 0.7]))
        keep = O.nms_greedy(boxes, scores, thr)
        order = np.argsort(-scores, kind="stable")
        assert list(keep) == [i for i in order if i in set(keep)]  # kept in descending stable order

        def iou(a, b):
            x1, y1 = max(a[0], b[0]), max(a[1], b[1])
            x2, y2 = min(a[2], b[2]), min(a[3], b[3])
            inter = np.float32(max(np.float32(0), x2 - x1)) * np.float32(max(np.float32(0), y2 - y1))
            return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)

        for a in range(len(keep)):
            for b in range(a + 1, len(keep)):
                assert not iou(boxes[keep[a]], boxes[keep[b]]) > np.float32(thr)
        rank = {int(i): r for r, i in enumerate(order)}
        for j in range(n):
            if j not in set(keep):
                assert any(rank[int(i)] < rank[j] and iou(boxes[i], boxes[j]) > np.float32(thr) for i in keep)


def test_e2e_vectors():
    import drone_yolo_amd as D

    g = golden("e2e.npz")
    for tag in ("n64", "n128", "sf_n64", "v8n320", "s640"):  # s640: the one full-size fixture, re-checked here too (0.3 s of CPU)
        m = meta(g, tag)
        d = load_yaml(m["yaml"], m["scale"], m["nc"])
        model = D.DetectionModel(dict(d), nc=m["nc"], verbose=False)
        tmpl = model.state_dict()
        assert sorted(tmpl.keys()) == [str(k) for k in g[f"{tag}__keys"]], f"{tag}: state-dict keys differ from the reference"
        assert [repr(tuple(tmpl[k].shape)) for k in sorted(tmpl)] == [str(s) for s in g[f"{tag}__shapes"]]
        assert sum(p.numel() for p in model.parameters()) == m["params"]
        sd = O.seeded_state_dict(tmpl, m["seed"], cls_bias=m["cls_bias"])
        b, h, w = m["shape"]
        x = torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(m["seed"]))
        with torch.no_grad():
            y, _ = O.forward(d, sd, x)
        if f"{tag}__y" in g.files:
            assert torch.allclose(y, torch.from_numpy(g[f"{tag}__y"]), atol=2e-3, rtol=1e-4), tag
        else:
            assert torch.allclose(y[:, :, ::37], torch.from_numpy(g[f"{tag}__y_sub"]), atol=2e-3, rtol=1e-4), tag
        det, idx = O.non_max_suppression(y, 0.25, 0.7, max_det=300, nc=m["nc"], return_index=True)
        assert [len(r) for r in det] == list(g[f"{tag}__n"]), tag
        assert np.array_equal(np.concatenate([i.numpy() for i in idx]), g[f"{tag}__det_idx"]), tag
        assert np.allclose(np.concatenate([r.numpy() for r in det]), g[f"{tag}__det"], atol=2e-3), tag


def test_augmented_inference_vectors():
    """predict(augment=True) of the REAL reference (DetectionModel._predict_augment, nn/tasks.py:347-383; tests/golden/aug.npz, oracle/make_golden.py::aug_vectors):
    the oracle's restatement — image pyramid 1 / 0.83 mirrored / 0.67, boxes scaled and mirrored back, the two clipped tails — gives the merged output and the rows."""
    import drone_yolo_amd as D

    g, ge = golden("aug.npz"), golden("e2e.npz")
    for tag in ("n128", "n64"):
        m = meta(ge, tag)
        d = load_yaml(m["yaml"], m["scale"], m["nc"])
        sd = O.seeded_state_dict(D.DetectionModel(dict(d), nc=m["nc"], verbose=False).state_dict(), m["seed"], cls_bias=m["cls_bias"])
        b, h, w = m["shape"]
        x = torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(m["seed"]))
        y = O.predict_augment(d, sd, x, fused=True)
        yref = torch.from_numpy(g[f"{tag}__y"])
        assert tuple(y.shape) == tuple(yref.shape)
        assert float((y - yref).abs().max()) <= 5e-5 * float(yref.abs().max())
        dets, idx = O.non_max_suppression(y, 0.25, 0.7, max_det=300, nc=m["nc"], return_index=True)
        assert [len(r) for r in dets] == [int(v) for v in g[f"{tag}__n"]]
        assert np.allclose(torch.cat(dets).numpy(), g[f"{tag}__det"], rtol=1e-4, atol=2e-3)
        assert np.array_equal(torch.cat(idx).numpy(), g[f"{tag}__det_idx"])
    # scale_img: the padded size follows ceil(h r / gs) gs, the pad value is 0.447
    z = O.scale_img(torch.zeros(1, 3, 128, 96), 0.67, gs=32)
    assert tuple(z.shape) == (1, 3, 96, 96) and float(z[0, 0, 90, 90]) == pytest.approx(0.447) and float(z[0, 0, 10, 10]) == 0.0


def test_bench_configuration_vectors():
    """tests/golden/big.npz::s640b4 / s640b4lo (BASELINE config 2 with the weights and inputs bench.py times, rows computed by the
    REAL reference): the oracle reproduces them here, and the package's seeded weight generator equals the oracle's.  The scale-l
    1280x1280 tiles and the scale-x 1536x1536 case of the same file cost minutes of CPU each: their oracle == reference
    check ran when make_golden.py wrote them (max |err| 0)."""
    import bench
    import drone_yolo_amd as D
    from drone_yolo_amd.utils import parity as PR

    g = golden("big.npz")
    for tag in ("s640b4", "s640bench"):
        m, x, exp_rows, exp_idx = PR.golden_case("big.npz", tag)
        assert x.shape == (4, 3, 640, 640)
        d = load_yaml(m["yaml"], m["scale"], m["nc"])
        d["yaml_file"] = m["yaml"].replace("yolov8", f"yolov8{m['scale']}")
        model = D.DetectionModel(dict(d), nc=m["nc"], verbose=False)
        sd = bench.fixture_weights(model, m)
        with torch.no_grad():
            y, _ = O.forward(d, sd, x[:2])
        assert torch.allclose(y[:, :, ::199], torch.from_numpy(g[f"{tag}__y_sub"][:2]), atol=2e-3, rtol=1e-4), tag
        det, idx = O.non_max_suppression(y, 0.25, 0.7, max_det=300, nc=m["nc"], return_index=True)
        for i in range(2):
            assert np.array_equal(idx[i].numpy(), exp_idx[i]), tag
            assert np.allclose(PR.clip_rows(det[i].numpy(), (640, 640)), exp_rows[i], atol=2e-3), tag
    tmpl = {"a.conv.weight": torch.zeros(8, 4, 3, 3), "a.bn.weight": torch.zeros(8), "a.bn.bias": torch.zeros(8), "a.bn.running_mean": torch.zeros(8),
            "a.bn.running_var": torch.zeros(8), "a.bn.num_batches_tracked": torch.zeros((), dtype=torch.long), "m.cv3.0.2.bias": torch.zeros(10),
            "m.dfl.conv.weight": torch.zeros(1, 16, 1, 1)}
    a, b = PR.seeded_state_dict(tmpl, 9, cls_bias=-2.0), O.seeded_state_dict(tmpl, 9, cls_bias=-2.0)
    assert all(torch.equal(a[k], b[k]) for k in tmpl)


def test_loss_stack_vectors():
    """bbox_iou(CIoU), TaskAlignedAssigner and v8DetectionLoss restatements against vectors captured from the
    real reference (oracle/make_golden.py::loss_vectors)."""
    from oracle import loss_oracle as LO

    g = golden("loss.npz")
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    assert torch.allclose(LO.bbox_ciou(t("ciou_b1"), t("ciou_b2")), t("ciou_out"), atol=1e-6)
    anc, st = O.make_anchors([(16, 16), (8, 8), (4, 4), (2, 2)], [4, 8, 16, 32])
    gt, gl = t("tal_gt"), t("tal_gl")
    mask_gt = (gt.sum(2, keepdim=True) > 0).float()
    tl, tb, ts, fg, gi = LO.task_aligned_assign(t("tal_scores"), t("tal_bboxes"), anc * st, gl, gt, mask_gt, topk=10, num_classes=10)
    assert torch.equal(fg, t("tal_fg")) and torch.equal(gi[fg], t("tal_gt_idx")[fg])
    assert torch.allclose(ts, t("tal_tscores"), atol=1e-6) and torch.allclose(tb[fg], t("tal_tbboxes")[fg])
    for tag in ("loss64", "loss160"):
        m = ast.literal_eval(str(g[f"{tag}_meta"]))
        gg = torch.Generator().manual_seed(m["seed"])
        feats = [torch.randn(m["bs"], 74, m["hw"] // s, m["hw"] // s, generator=gg) * 1.5 for s in (4, 8, 16, 32)]
        labels = LO.synthetic_labels(m["bs"], m["seed"], n_mean=m["n_mean"])
        total, items, asg = LO.v8_detection_loss(feats, labels, [4.0, 8.0, 16.0, 32.0], 10, return_assign=True)
        assert torch.allclose(items, t(f"{tag}_items"), rtol=2e-5, atol=1e-5), tag
        assert abs(float(total) - float(g[f"{tag}_total"])) <= 2e-5 * abs(float(g[f"{tag}_total"])), tag
        assert torch.equal(asg["fg_mask"], t(f"{tag}_fg")) and torch.equal(asg["target_gt_idx"][asg["fg_mask"]], t(f"{tag}_gt_idx")[asg["fg_mask"]])


def test_train_step_vectors():
    """oracle/train_oracle.py (train-mode forward, loss, autograd) against the gradients captured from the REAL
    reference's module.train() + v8DetectionLoss + backward (oracle/make_golden.py::train_vectors)."""
    import drone_yolo_amd as D
    from oracle import loss_oracle as LO
    from oracle import train_oracle as TO

    g = golden("train.npz")
    for tag in ("tn64",):
        m = meta(g, tag)
        d = load_yaml(m["yaml"], m["scale"], m["nc"])
        tmpl = D.DetectionModel(dict(d), nc=m["nc"], verbose=False).state_dict()
        assert sorted(tmpl.keys()) == [str(k) for k in g[f"{tag}__keys"]]
        sd = O.seeded_state_dict(tmpl, m["seed"], cls_bias=m["cls_bias"])
        b, h, w = m["shape"]
        img = torch.randint(0, 256, (b, 3, h, w), generator=torch.Generator().manual_seed(m["seed"]), dtype=torch.uint8)
        labels = LO.synthetic_labels(b, m["seed"], n_mean=m["n_mean"])
        total, items, grads, _ = TO.loss_and_grads(d, sd, img, labels)
        assert abs(float(total) - float(g[f"{tag}__total"])) <= 2e-5 * abs(float(g[f"{tag}__total"]))
        assert torch.allclose(items, torch.from_numpy(g[f"{tag}__items"]), rtol=2e-5)
        keys = [str(k) for k in g[f"{tag}__grad_keys"]]
        norms = np.array([float(grads[k].double().norm()) for k in keys])
        assert np.allclose(norms, g[f"{tag}__grad_norm"], rtol=2e-4, atol=1e-7)
        for f in g.files:
            if f.startswith(f"{tag}__grad::"):
                k = f.split("::", 1)[1]
                ref = torch.from_numpy(g[f])
                assert float((grads[k] - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-9, k


def test_letterbox_oracle_resize_sane():
    """The restated 8-bit bilinear stays within 1 LSB of a float bilinear (it is the same interpolation, fixed point)."""
    from oracle import letterbox_oracle as LB

    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (211, 173, 3), dtype=np.uint8)
    r = LB.resize_linear_u8(img, 301, 97).astype(np.float32)
    import torch.nn.functional as F

    t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
    f = F.interpolate(t, size=(97, 301), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(r - f).max() <= 1.0
