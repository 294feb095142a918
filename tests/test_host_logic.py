"""CPU: host-side logic of the package — YAML parsing, graph planning, weight folding/packing rules,
sharding, and the guarantee that nothing silently falls back to CPU compute."""
import math

import numpy as np
import pytest
import torch

import drone_yolo_amd as D
from drone_yolo_amd.nn import tasks
from drone_yolo_amd.nn.modules import C2f, Concat, Conv, Detect, RepVGGBlock, Upsample
from drone_yolo_amd.nn.modules.conv import fold_conv_bn
from oracle import drone_yolo_oracle as O


def test_yaml_scale_and_param_counts():
    # parameter counts probed from the reference
    for name, nc, params in (("yolov8n-p2-repvgg.yaml", 10, 2_972_360), ("yolov8s-p2-repvgg.yaml", 10, 10_815_576),
                             ("yolov8s-p2-repvgg-sf.yaml", 10, 10_839_320), ("yolov8n.yaml", 80, 3_157_200)):
        m = D.DetectionModel(name, nc=nc, verbose=False)
        assert sum(p.numel() for p in m.parameters()) == params, name
    m = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False)
    assert m.stride.tolist() == [4.0, 8.0, 16.0, 32.0]
    assert isinstance(m.model[1], RepVGGBlock) and m.model[1].rbr_dense.conv.stride == (2, 2)
    assert m.model[-1].legacy is True and m.model[-1].no == 74
    # Detect.bias_init (head.py:133-144)
    det = m.model[-1]
    assert torch.allclose(det.cv2[0][-1].bias, torch.ones(64))
    assert math.isclose(float(det.cv3[1][-1].bias[0]), math.log(5 / 10 / (640 / 8) ** 2), rel_tol=1e-6)
    # BN constants (torch_utils.py:423-433)
    assert m.model[0].bn.eps == 1e-3 and m.model[0].bn.momentum == 0.03


def test_state_dict_keys_follow_reference_names():
    m = D.DetectionModel("yolov8n-p2-repvgg.yaml", nc=10, verbose=False)
    keys = set(m.state_dict())
    for k in ("model.0.conv.weight", "model.0.bn.running_mean", "model.1.rbr_dense.conv.weight", "model.1.rbr_1x1.bn.running_var",
              "model.2.m.0.cv1.conv.weight", "model.9.cv2.bn.bias", "model.28.cv2.0.0.conv.weight", "model.28.cv3.3.2.bias",
              "model.28.dfl.conv.weight"):
        assert k in keys, k
    assert torch.equal(m.state_dict()["model.28.dfl.conv.weight"].flatten(), torch.arange(16.0))


def test_graph_plan_folds_upsample_concat_and_places_producers():
    m = D.DetectionModel("yolov8s-p2-repvgg.yaml", nc=10, verbose=False)
    m._plan_graph()
    assert m._virtual == {11: (9, 6), 14: (12, 4), 17: (15, 2)}
    assert m._skip == {10, 11, 13, 14, 16, 17}
    assert m._place == {19: (20, 0), 15: (20, 64), 22: (23, 0), 12: (23, 128), 25: (26, 0), 9: (26, 256)}
    sf = D.DetectionModel("yolov8s-p2-repvgg-sf.yaml", nc=10, verbose=False)
    sf._plan_graph()
    assert 12 not in sf._virtual and sf._place[11] == (12, 0) and sf._place[6] == (12, 64) and sf._place[10] == (12, 320)


def test_fold_rules_match_oracle():
    torch.manual_seed(0)
    c = Conv(8, 16, 3, 2)
    sd = O.seeded_state_dict(c.state_dict(), 5)
    c.load_state_dict(sd)
    tasks.initialize_weights(c)
    w, b = fold_conv_bn(c.conv.weight, None, c.bn)
    wo, bo = O.fuse_conv_bn(sd["conv.weight"], {f"bn.{k[3:]}": v for k, v in sd.items() if k.startswith("bn.")}, "bn")
    assert torch.allclose(w, wo, atol=1e-6) and torch.allclose(b, bo, atol=1e-6)
    for (c1, c2, s) in ((8, 16, 2), (8, 8, 1)):
        r = RepVGGBlock(c1, c2, 3, s)
        sd = O.seeded_state_dict(r.state_dict(), 6)
        r.load_state_dict(sd)
        tasks.initialize_weights(r)
        k, bb = r.get_equivalent_kernel_bias()
        ko, bo = O.repvgg_equivalent({f"m.{k_}": v for k_, v in sd.items()}, "m", s == 1, c1)
        assert torch.allclose(k, ko, atol=1e-6) and torch.allclose(bb, bo, atol=1e-6)
        r.switch_to_deploy()
        assert r.deploy and torch.allclose(r.rbr_reparam.weight, ko, atol=1e-6)


def test_packed_weight_layout():
    """Row co = (r, q, c)-ordered taps, zero padding to k_pad / cout_pad (include/dyolo.h)."""
    from drone_yolo_amd import hip_ops as H

    w = torch.arange(10 * 8 * 3 * 3, dtype=torch.float32).view(10, 8, 3, 3) / 100
    pc = H.PackedConv(w, torch.arange(10.0), 1, 1, 1, True, torch.float32, "cpu")
    assert tuple(pc.w.shape) == (64, 96) and pc.k_pad == 96 and pc.cout_pad == 64
    assert pc.w[3, (1 * 3 + 2) * 8 + 5] == w[3, 5, 1, 2]
    assert float(pc.w[10:].abs().sum()) == 0 and float(pc.w[:, 72:].abs().sum()) == 0
    assert torch.equal(pc.b[:10], torch.arange(10.0)) and float(pc.b[10:].abs().sum()) == 0
    pc16 = H.PackedConv(w, torch.zeros(10), 1, 1, 1, True, torch.bfloat16, "cpu")
    assert tuple(pc16.w.shape) == (64, 128) and pc16.w.dtype == torch.bfloat16


def test_no_cpu_fallback():
    m = D.DetectionModel("yolov8n-p2-repvgg.yaml", nc=10, verbose=False).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.model[0](torch.zeros(1, 8, 32, 32).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2))
    from drone_yolo_amd.utils.torch_utils import select_device

    with pytest.raises(RuntimeError):
        select_device("cpu")
    with pytest.raises(NotImplementedError):  # dataset YAMLs / image folders: the loader side is out of scope
        D.YOLO("yolov8n-p2-repvgg.yaml").train(data="x.yaml")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            D.YOLO("yolov8n-p2-repvgg.yaml").train(data="synthetic:4", epochs=1)
    m.train()  # a module called on its own in training mode runs the device training forward (r04) — and has no CPU fallback either
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.model[0](torch.zeros(1, 8, 8, 8).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2))
    with pytest.raises(NotImplementedError):  # eval-path launch options have no training form
        m.model[0](torch.zeros(1, 8, 8, 8), residual=torch.zeros(1))


def test_view_params_and_alloc():
    from drone_yolo_amd import hip_ops as H

    t = H.alloc_nhwc(2, 24, 5, 7, torch.bfloat16, "cpu")
    assert tuple(t.shape) == (2, 24, 5, 7) and t.stride() == (5 * 7 * 24, 1, 7 * 24, 24)
    p, ld = H.view_params(t[:, 8:16])
    assert ld == 24 and p == t.data_ptr() + 8 * 2
    with pytest.raises(ValueError):
        H.view_params(torch.zeros(2, 24, 5, 7))  # NCHW-contiguous is not an NHWC view
    t2 = H.alloc_nhwc(1, 10, 4, 4, torch.float32, "cpu", ld=12)
    assert H.view_params(t2)[1] == 12


def test_shard_range_covers_batch():
    from drone_yolo_amd.parallel import shard_range

    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


def test_ops_host_helpers():
    from drone_yolo_amd.utils import ops

    b = torch.tensor([[10.0, 20, 4, 6]])
    assert torch.equal(ops.xywh2xyxy(b), torch.tensor([[8.0, 17, 12, 23]]))
    assert ops.make_divisible(65, 8) == 72
    bb = torch.tensor([[-5.0, 10, 700, 500], [30, 40, 50, 60]])
    assert torch.allclose(ops.scale_boxes((384, 640), bb.clone(), (480, 800)), O.scale_boxes((384, 640), bb.clone(), (480, 800)))


def test_reference_checkpoint_reader():
    """A checkpoint pickled by the REAL reference (module graph under 'ema', fp16; oracle/make_golden.py::checkpoint_fixture)
    is read without the ultralytics package: same keys/shapes/parameter count, and the oracle forward on the loaded
    weights reproduces the output the reference computed from that checkpoint."""
    import os
    import pickle

    import numpy as np
    import pytest
    import torch

    from drone_yolo_amd.nn.checkpoint import RefUnpickler, load_reference_checkpoint, read_reference_checkpoint
    from oracle import drone_yolo_oracle as O

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "ref_checkpoint_t.pt")
    g = np.load(os.path.join(root, "tests", "golden", "ref_checkpoint_t.npz"))
    yaml_d, sd, meta = read_reference_checkpoint(path)
    assert yaml_d["scale"] == "t" and yaml_d["nc"] == 10 and meta["epoch"] == 3 and meta["names"][3] == "class3"
    assert all(v.dtype == torch.float32 for v in sd.values() if v.is_floating_point())
    model, _ = load_reference_checkpoint(path)
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"])
    assert model.names[3] == "class3"
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        y, _ = O.forward(dict(yaml_d), {k: v for k, v in model.state_dict().items()}, x, fused=False)
    assert torch.allclose(y, torch.from_numpy(g["y"]), atol=2e-3, rtol=1e-4)
    # the unpickler refuses anything outside torch / collections / numpy / the mapped reference classes
    import io

    class _E:
        def __reduce__(self):
            return (os.system, ("true",))
    with pytest.raises(pickle.UnpicklingError):
        RefUnpickler(io.BytesIO(pickle.dumps(_E()))).load()


def test_checkpoint_unpickler_refuses_bypass_payloads():
    """a prefix allow-list let protocol-4 dotted globals walk out of torch / numpy
    (('torch', 'serialization.os.getcwd') resolved to os.getcwd).  The allow-list is exact now: each of the three verified
    payloads, builtins.getattr and types.FunctionType must raise UnpicklingError without running anything."""
    import io
    import pickle

    import pytest

    from drone_yolo_amd.nn.checkpoint import RefUnpickler

    def stack_global(module: str, name: str, call: bool = True) -> bytes:
        def s(x):
            b = x.encode()
            return b"\x8c" + bytes([len(b)]) + b
        return b"\x80\x04" + s(module) + s(name) + b"\x93" + (b")R" if call else b"") + b"."

    for module, name in [("torch", "serialization.os.getcwd"), ("numpy", "testing._private.utils.runstring"), ("types", "FunctionType"),
                         ("builtins", "getattr"), ("torch", "load"), ("torch.serialization", "os"), ("copyreg", "_reconstructor"),
                         ("torch.nn.modules.module", "torch"), ("torch.nn.modules.conv", "F"), ("argparse", "ArgumentParser"), ("os", "system")]:
        with pytest.raises(pickle.UnpicklingError):
            RefUnpickler(io.BytesIO(stack_global(module, name))).load()
    # what a checkpoint legitimately needs still resolves
    import collections

    assert RefUnpickler(io.BytesIO(stack_global("collections", "OrderedDict"))).load() == collections.OrderedDict()
    assert RefUnpickler(io.BytesIO(stack_global("torch.nn.modules.conv", "Conv2d", call=False))).load() is __import__("torch").nn.Conv2d


def test_optimizer_schedule_matches_reference_loop():
    """warm-up and optimizer selection against the reference's loop restated around real torch.optim
    objects (oracle/train_oracle.py::reference_schedule; trainer.py:362-377, 784-793): per batch the three group lrs, SGD
    momentum (AdamW's beta1 untouched), the accumulate ramp and which batches step the optimizer — with warmup_epochs > 0,
    for explicit SGD / AdamW and for optimizer='auto' on both sides of its 10,000-iteration rule."""
    from drone_yolo_amd.engine.trainer import OptimSchedule, get_cfg, resolve_optimizer
    from oracle import train_oracle as TO

    for opt, batch, iters in [("SGD", 16, 50), ("AdamW", 16, 50), ("auto", 16, 50), ("auto", 64, 20000), ("SGD", 64, 50)]:
        a = get_cfg(dict(optimizer=opt, batch=batch, epochs=4, warmup_epochs=3.0))
        nb, epochs = 40, 4
        name, lr0, mom, wbl = resolve_optimizer(a, 10, iters)
        ref_name, ref_decay, rows = TO.reference_schedule(a, nb, batch, epochs, iters, nc=10)
        assert name == ref_name
        s = OptimSchedule(a, name, lr0, mom, wbl, batch, epochs)
        assert abs(a["weight_decay"] * batch * s.accumulate / a["nbs"] - ref_decay) < 1e-15
        last = -1
        for epoch in range(epochs):
            s.scheduler_step(epoch)
            for i in range(nb):
                ni = i + nb * epoch
                s.warmup(ni, epoch, nb)
                stepped = ni - last >= s.accumulate
                if stepped:
                    last = ni
                r_ni, r_lrs, r_mom, r_b1, r_acc, r_step = rows[ni]
                assert r_ni == ni and stepped == r_step and s.accumulate == r_acc, (opt, ni)
                # reference group order: biases, decay weights, norm weights
                got = [s.cur_lrs[2], s.cur_lrs[0], s.cur_lrs[1]]
                assert all(abs(g - r) <= 1e-12 + 1e-9 * abs(r) for g, r in zip(got, r_lrs)), (opt, ni, got, r_lrs)
                if name == "SGD":
                    assert abs(s.cur_momentum - r_mom) < 1e-12, (opt, ni)
                else:
                    assert r_mom is None and abs(s.momentum - r_b1) < 1e-12  # beta1 never warmed up


def test_tensor_loader_is_a_distributed_sampler():
    """TensorLoader == torch's DistributedSampler (data/build.py:144) + the reference's collate: same indices per rank and
    epoch, ranks cover the padded set exactly once, labels re-indexed by position in the batch."""
    from torch.utils.data.distributed import DistributedSampler

    from drone_yolo_amd.engine.trainer import TensorLoader, synthetic_dataset

    d = synthetic_dataset(11, 16, seed=3)
    for epoch in (0, 2):
        seen = []
        for rank in range(2):
            ld = TensorLoader(d, 4, rank, 2, seed=5)
            ld.set_epoch(epoch)
            ds = DistributedSampler(range(11), num_replicas=2, rank=rank, shuffle=True, seed=5)
            ds.set_epoch(epoch)
            assert ld.indices() == list(ds)
            assert len(ld) == 2
            seen += ld.indices()
            batches = list(ld)
            take = ld.indices()[:4]
            b = batches[0]
            assert torch.equal(b["img"], d["img"][take])
            for j, i in enumerate(take):
                m_src, m_dst = d["batch_idx"] == i, b["batch_idx"] == j
                assert torch.equal(b["bboxes"][m_dst], d["bboxes"][m_src]) and torch.equal(b["cls"][m_dst], d["cls"][m_src])
        assert sorted(set(seen)) == list(range(11)) and len(seen) == 12


def test_requested_device_ids_reach_the_ranks(monkeypatch):
    """``device="2,3"`` must train on GPUs 2 and 3 (select_device exports CUDA_VISIBLE_DEVICES=device, reference
    utils/torch_utils.py:183), not on 0..N-1: the launcher's rank environment carries HIP_VISIBLE_DEVICES = the requested list,
    mapped through an outer visibility list, validated against the device count without initialising HIP."""
    from drone_yolo_amd.utils import dist as DI

    monkeypatch.setattr(DI, "visible_gpu_count", lambda: 8)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    env = DI.rank_env(DI.visible_device_env([2, 3]))
    assert env["HIP_VISIBLE_DEVICES"] == "2,3" and "CUDA_VISIBLE_DEVICES" not in env
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert DI.rank_env(DI.visible_device_env([4, 5, 6, 7]))["HIP_VISIBLE_DEVICES"] == "4,5,6,7"
    with pytest.raises(RuntimeError, match="not among"):
        DI.visible_device_env([2, 9])
    with pytest.raises(RuntimeError, match="duplicates"):
        DI.visible_device_env([1, 1])
    # an outer list (this process sees 4 GPUs that are physical 4..7): positions map through it and the stale CUDA_ list is dropped
    monkeypatch.setattr(DI, "visible_gpu_count", lambda: 4)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "0,1")
    env = DI.rank_env(DI.visible_device_env([1, 3]))
    assert env["HIP_VISIBLE_DEVICES"] == "5,7" and "CUDA_VISIBLE_DEVICES" not in env


def test_optimizer_groups_enumerate_frozen_parameters_like_the_reference():
    """build_optimizer (reference trainer.py:795-819) does not filter on requires_grad, so its decay group holds the
    frozen ``model.N.dfl.conv.weight`` and every later index counts it; the state-dict enumeration must do the same (the flat
    training buffers still hold trainable parameters only)."""
    from drone_yolo_amd.engine.trainer import param_group_names

    m = D.DetectionModel("yolov8n-p2-repvgg.yaml", nc=10, verbose=False)
    for k, v in m.named_parameters():
        v.requires_grad_(".dfl" not in k)
    g0, g1, g2 = param_group_names(m)
    f0, f1, f2 = param_group_names(m, include_frozen=True)
    assert "model.28.dfl.conv.weight" not in g0 and "model.28.dfl.conv.weight" in f0
    assert len(f0) == len(g0) + 1 and f1 == g1 and f2 == g2
    # the reference's own enumeration, restated: every module's direct parameters, bias / norm weight / other
    ref0, ref1, ref2 = [], [], []
    bn = tuple(v for k, v in torch.nn.__dict__.items() if "Norm" in k)
    for mn, mod in m.named_modules():
        for pn, p in mod.named_parameters(recurse=False):
            full = f"{mn}.{pn}" if mn else pn
            (ref2 if "bias" in full else ref1 if isinstance(mod, bn) else ref0).append(full)
    assert (f0, f1, f2) == (ref0, ref1, ref2) and len(ref0) + len(ref1) + len(ref2) == len(list(m.parameters()))


def test_lazy_head_seed_refuses_unconsumed_scales():
    """nn/autograd_ops.lazy_head_seed: a head gradient handed on without its backward seed must be consumed (HeadTail.backward pops
    it); anything left at the end of the trainer's backward is an error, not a silently unscaled gradient."""
    import pytest

    from drone_yolo_amd.nn import autograd_ops as A

    with A.lazy_head_seed():
        A.LAZY_SEED[1234] = object()
        assert A.LAZY_SEED.pop(1234, None) is not None  # consumed: fine
    with pytest.raises(RuntimeError, match="without their scale"):
        with A.lazy_head_seed():
            A.LAZY_SEED[5678] = object()
    assert not A.LAZY_SEED and not A._LAZY[0]


def test_validation_metrics_match_the_reference_functions():
    """utils/metrics.py (the host half of the validation step: box_iou, match_predictions, ap_per_class, fitness) against the REAL
    reference's functions on synthetic detections (tests/golden/val_metrics.npz, oracle/make_golden.py::val_metric_vectors):
    the matching is an exact boolean matrix; AP / precision / recall to 1e-12 (same numpy arithmetic in the same order)."""
    import numpy as np

    from drone_yolo_amd.utils import metrics as M
    from tests._util import golden

    g = golden("val_metrics.npz")
    iouv = np.linspace(0.5, 0.95, 10)
    for i in range(int(g["n_img"])):
        gtb, gtc, db, dc = g[f"img{i}_gtb"], g[f"img{i}_gtc"], g[f"img{i}_db"], g[f"img{i}_dc"]
        iou = M.box_iou(gtb, db)
        assert iou.shape == g[f"img{i}_iou"].shape and np.allclose(iou, g[f"img{i}_iou"], rtol=1e-6, atol=1e-7)
        if len(gtc) and len(dc):
            tp = M.match_predictions(dc, gtc, g[f"img{i}_iou"], iouv)
            assert np.array_equal(tp, g[f"img{i}_tp"]), i
    res = M.ap_per_class(g["tp"], g["conf"], g["pred_cls"], g["target_cls"])
    for got, key in zip(res, ("ap_tp", "ap_fp", "ap_p", "ap_r", "ap_f1", "ap_ap", "ap_classes")):
        assert np.allclose(got, g[key], rtol=0, atol=1e-12), key
    m = M.DetMetrics()
    m.process(g["tp"], g["conf"], g["pred_cls"], g["target_cls"])
    assert np.allclose(m.mean_results(), g["mean_results"], atol=1e-12) and abs(m.fitness - float(g["fitness"])) < 1e-12
    assert set(m.results_dict) == {"metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)", "fitness"}


def test_check_cfg_types_and_ranges():
    """engine/trainer.py::check_cfg against the rules of the reference's check_cfg (cfg/__init__.py:324-395): float / fraction / int / bool keys, None = unset."""
    from drone_yolo_amd.engine.trainer import check_cfg, get_cfg

    cfg = get_cfg(dict(epochs=3, lr0=0.01, batch=16, time=None, save=True))
    assert cfg["epochs"] == 3 and cfg["patience"] == 100 and cfg["resume"] is False
    for bad, exc in ((dict(momentum=1.2), ValueError), (dict(epochs=2.5), TypeError), (dict(save="true"), TypeError), (dict(lr0="0.1"), TypeError), (dict(conf=-0.1), ValueError),
                     (dict(warmup_epochs="3"), TypeError), (dict(nbs=True), TypeError)):
        with pytest.raises(exc):
            get_cfg(bad)
    soft = dict(epochs=2.0, save=1, lr0="0.1", time=None)
    check_cfg(soft, hard=False)
    assert soft == dict(epochs=2, save=True, lr0=0.1, time=None) and isinstance(soft["epochs"], int)
    with pytest.raises(KeyError):
        get_cfg(dict(not_an_argument=1))


def test_results_text_forms(tmp_path):
    """engine/results.py: verbose / summary / to_json / save_txt / indexing of a detection result (reference results.py:633-940), on hand-made rows."""
    import json

    from drone_yolo_amd.engine.results import Results

    rows = torch.tensor([[10.0, 20.0, 110.0, 220.0, 0.9, 1.0], [0.0, 0.0, 50.0, 40.0, 0.6, 0.0], [5.0, 5.0, 25.0, 45.0, 0.3, 1.0]])
    r = Results(torch.zeros(3, 400, 200), "a.jpg", {0: "person", 1: "car"}, boxes=rows, orig_shape=(400, 200))
    assert r.verbose() == "1 person, 2 cars, " and len(r) == 3 and len(r[0]) == 1 and len(r[rows[:, 4] > 0.5]) == 2
    s = r.summary()
    assert s[0] == {"name": "car", "class": 1, "confidence": 0.9, "box": {"x1": 10.0, "y1": 20.0, "x2": 110.0, "y2": 220.0}}
    assert r.summary(normalize=True)[0]["box"] == {"x1": 0.05, "y1": 0.05, "x2": 0.55, "y2": 0.55}
    assert json.loads(r.to_json())[1]["name"] == "person" and r.tojson() == r.to_json()
    out = r.save_txt(tmp_path / "labels" / "a.txt", save_conf=True)
    lines = open(out).read().splitlines()
    assert lines[0] == "1 0.3 0.3 0.5 0.5 0.9" and lines[1] == "0 0.125 0.05 0.25 0.1 0.6" and len(lines) == 3
    empty = Results(torch.zeros(3, 8, 8), "b.jpg", {0: "x"}, boxes=torch.zeros(0, 6), orig_shape=(8, 8))
    assert empty.verbose() == "(no detections), " and empty.summary() == [] and isinstance(r.numpy().boxes.data, np.ndarray)


def test_image_file_sources_are_listed_and_decoded_like_the_reference_loader(tmp_path):
    """Host glue in front of the path (reference data/loaders.py:284-420 LoadImagesAndVideos, :451-500 LoadPilAndNumpy): a file, a directory, a glob, a
    .txt list or a list of them -> the image files in sorted order; decoding -> contiguous HWC BGR uint8 (RGB reversed), greyscale / RGBA converted."""
    import numpy as np
    from PIL import Image

    from drone_yolo_amd.engine.predictor import DetectionPredictor as P

    rng = np.random.default_rng(0)
    arrs = {}
    for i, (h, w) in enumerate(((40, 50), (33, 47), (64, 64))):
        arrs[f"b{i}.png"] = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(arrs[f"b{i}.png"]).save(tmp_path / f"b{i}.png")
    Image.fromarray(rng.integers(0, 256, (20, 30), dtype=np.uint8)).save(tmp_path / "grey.bmp")
    (tmp_path / "notes.md").write_text("not an image")
    (tmp_path / "list.txt").write_text("b2.png\nb0.png\n")
    base = lambda fs: [f.rsplit("/", 1)[-1] for f in fs]  # noqa: E731
    assert base(P.list_image_files(str(tmp_path))) == ["b0.png", "b1.png", "b2.png", "grey.bmp"]  # the directory's *.*, sorted; the .md / .txt skipped
    assert base(P.list_image_files(str(tmp_path / "*.png"))) == ["b0.png", "b1.png", "b2.png"]
    assert base(P.list_image_files(str(tmp_path / "list.txt"))) == ["b0.png", "b2.png"]  # relative to the list's directory, sorted as the reference sorts
    assert base(P.list_image_files([tmp_path / "b1.png", str(tmp_path / "b0.png")])) == ["b0.png", "b1.png"]
    with pytest.raises(FileNotFoundError):
        P.list_image_files(str(tmp_path / "nothing.png"))
    with pytest.raises(NotImplementedError):
        P.list_image_files(str(tmp_path / "notes.md"))
    a = P.decode_image(str(tmp_path / "b1.png"))
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"] and np.array_equal(a, arrs["b1.png"][:, :, ::-1])  # BGR
    gimg = P.decode_image(str(tmp_path / "grey.bmp"))
    assert gimg.shape == (20, 30, 3) and np.array_equal(gimg[..., 0], gimg[..., 2])
    rgba = Image.fromarray(rng.integers(0, 256, (8, 9, 4), dtype=np.uint8), "RGBA")
    assert P.decode_image(rgba).shape == (8, 9, 3)
