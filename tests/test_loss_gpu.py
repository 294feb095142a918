"""GPU parity of dy_detection_loss (TaskAlignedAssigner + BCE/CIoU/DFL) against the oracle restatement of the
reference's v8DetectionLoss, on random head outputs and VisDrone-like synthetic labels.

Tolerance: loss terms to 2e-4 relative (fp32 sums in a different order; double accumulators on the device); the
foreground set and its box assignment must be identical except for anchors whose alignment metric is exactly 0
(torch.topk returns such ties in unspecified order; they carry target score 0 and change no loss term)."""
import ast

import pytest
import torch

from drone_yolo_amd import hip_ops as H
from oracle import loss_oracle as LO
from tests._util import golden

pytestmark = pytest.mark.gpu
STRIDES = [4.0, 8.0, 16.0, 32.0]


def _dev_feats(feats, device):
    return [f.permute(0, 2, 3, 1).contiguous().to(device).permute(0, 3, 1, 2) for f in feats]


def _gt(labels, bs, hw):
    scale = torch.tensor([hw, hw, hw, hw], dtype=torch.float32)
    return LO.preprocess_targets(labels["batch_idx"].view(-1), labels["cls"].view(-1), labels["bboxes"], bs, scale)


@pytest.mark.parametrize("bs,hw,seed,n_mean", [(2, 64, 7, 6.0), (3, 160, 8, 14.0), (4, 320, 9, 40.0), (2, 160, 10, 0.01)])
def test_loss_matches_oracle(bs, hw, seed, n_mean, device):
    gg = torch.Generator().manual_seed(seed)
    feats = [torch.randn(bs, 74, hw // int(s), hw // int(s), generator=gg) * 1.5 for s in STRIDES]
    labels = LO.synthetic_labels(bs, seed, n_mean=n_mean)
    total, items, asg = LO.v8_detection_loss(feats, labels, STRIDES, 10, return_assign=True)
    out, owner = H.detection_loss(_dev_feats(feats, device), _gt(labels, bs, hw), STRIDES, 10, want_owner=True)
    torch.cuda.synchronize()
    out, owner = out.cpu(), owner.cpu().long()
    assert torch.allclose(out[:3], items, rtol=2e-4, atol=1e-5), (out, items)
    assert abs(float(out[3]) - float(total.detach())) <= 2e-4 * abs(float(total.detach())) + 1e-5
    # assignment: positive-target anchors identical, same ground-truth box
    pos = asg["target_scores"].sum(-1) > 0
    fg_dev = owner >= 0
    assert torch.equal(fg_dev & pos, pos), "device misses foreground anchors the reference has"
    assert not bool((fg_dev & ~asg["fg_mask"]).any()), "device has foreground anchors the reference lacks"
    assert torch.equal(owner[pos], asg["target_gt_idx"][pos])


def test_loss_golden_vectors(device):
    """Against the numbers captured from the REAL reference (tests/golden/loss.npz)."""
    g = golden("loss.npz")
    for tag in ("loss64", "loss160"):
        m = ast.literal_eval(str(g[f"{tag}_meta"]))
        gg = torch.Generator().manual_seed(m["seed"])
        feats = [torch.randn(m["bs"], 74, m["hw"] // int(s), m["hw"] // int(s), generator=gg) * 1.5 for s in STRIDES]
        labels = LO.synthetic_labels(m["bs"], m["seed"], n_mean=m["n_mean"])
        out, owner = H.detection_loss(_dev_feats(feats, device), _gt(labels, m["bs"], m["hw"]), STRIDES, 10, want_owner=True)
        torch.cuda.synchronize()
        assert torch.allclose(out[:3].cpu(), torch.from_numpy(g[f"{tag}_items"]), rtol=2e-4, atol=1e-5), tag
        assert abs(float(out[3]) - float(g[f"{tag}_total"])) <= 2e-4 * abs(float(g[f"{tag}_total"]))


def test_loss_class_api_and_empty_labels(device):
    import drone_yolo_amd as D
    from drone_yolo_amd.utils.loss import v8DetectionLoss

    model = D.DetectionModel("yolov8n-p2-repvgg.yaml", nc=10, verbose=False)
    crit = v8DetectionLoss(model)
    gg = torch.Generator().manual_seed(3)
    feats = [torch.randn(2, 74, 160 // int(s), 160 // int(s), generator=gg) for s in STRIDES]
    labels = LO.synthetic_labels(2, 3, n_mean=10.0)
    loss, items = crit(_dev_feats(feats, device), labels)
    total, ref_items = LO.v8_detection_loss(feats, labels, STRIDES, 10)
    assert torch.allclose(items.cpu(), ref_items, rtol=2e-4, atol=1e-5) and abs(float(loss.detach()) - float(total)) <= 2e-4 * float(total)
    empty = {"batch_idx": torch.zeros(0), "cls": torch.zeros(0, 1), "bboxes": torch.zeros(0, 4)}
    loss0, items0 = crit(_dev_feats(feats, device), empty)
    t0, i0 = LO.v8_detection_loss(feats, empty, STRIDES, 10)
    assert torch.allclose(items0.cpu(), i0, rtol=2e-4, atol=1e-5) and float(items0[0]) == 0.0 and float(items0[2]) == 0.0


@pytest.mark.parametrize("bs,hw,seed,n_mean", [(2, 64, 7, 6.0), (3, 160, 8, 14.0), (2, 320, 11, 30.0)])
def test_loss_gradient_matches_autograd(bs, hw, seed, n_mean, device):
    """d(loss.sum() * B)/d head outputs from the device against autograd through the oracle (itself equal to the
    reference's gradient to the last bit: oracle/make_golden.py loss_vectors), tolerance 1e-4 of the largest entry."""
    gg = torch.Generator().manual_seed(seed)
    feats = [(torch.randn(bs, 74, hw // int(s), hw // int(s), generator=gg) * 1.5).requires_grad_(True) for s in STRIDES]
    labels = LO.synthetic_labels(bs, seed, n_mean=n_mean)
    total, _ = LO.v8_detection_loss(feats, labels, STRIDES, 10)
    total.backward()
    out, _, grads = H.detection_loss(_dev_feats([f.detach() for f in feats], device), _gt(labels, bs, hw), STRIDES, 10, want_grad=True)
    torch.cuda.synchronize()
    assert abs(float(out[3]) - float(total.detach())) <= 2e-4 * abs(float(total.detach()))
    for f, gd in zip(feats, grads):
        ref = f.grad
        err = float((gd.cpu() - ref).abs().max())
        assert err <= 1e-4 * float(ref.abs().max()) + 1e-7, (err, float(ref.abs().max()))


def test_loss_gradient_golden_reference(device):
    """The small case's full gradient and the big case's per-level sums as captured from the REAL reference."""
    g = golden("loss.npz")
    for tag in ("loss64", "loss160"):
        m = ast.literal_eval(str(g[f"{tag}_meta"]))
        gg = torch.Generator().manual_seed(m["seed"])
        feats = [torch.randn(m["bs"], 74, m["hw"] // int(s), m["hw"] // int(s), generator=gg) * 1.5 for s in STRIDES]
        labels = LO.synthetic_labels(m["bs"], m["seed"], n_mean=m["n_mean"])
        _, _, grads = H.detection_loss(_dev_feats(feats, device), _gt(labels, m["bs"], m["hw"]), STRIDES, 10, want_grad=True)
        torch.cuda.synchronize()
        for li, gd in enumerate(grads):
            gd = gd.cpu()
            assert abs(float(gd.abs().sum()) - float(g[f"{tag}_grad_abs_sum"][li])) <= 2e-4 * float(g[f"{tag}_grad_abs_sum"][li])
            if tag == "loss64":
                ref = torch.from_numpy(g[f"{tag}_grad{li}"])
                assert float((gd - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
