"""Faster R-CNN path: FasterRCNNEngine (ResNet-FPN + RPN head on the MFMA kernels) against oracle/retina_oracle.py:frcnn_forward and
the FasterRCNN module mirror end to end."""
import numpy as np
import pytest

from oracle import detrand
from oracle import retina_oracle as ro

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

PX, BS, SEED = 128, 2, 7300


def dev():
    return torch.device("cuda:0")


def nchw(a):
    return a.buf.float().permute(0, 3, 1, 2).cpu()


def rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.fixture(scope="module")
def setup():
    from object_detectors_amd.tvision.engine import FasterRCNNEngine
    sd = ro.det_state(SEED, keys=ro.frcnn_state_keys())
    eng = FasterRCNNEngine(3, device=dev(), seed=0)
    eng.load_reference_state_dict(sd)
    x = torch.from_numpy(detrand.uniform(4242, (BS, 3, PX, PX), 0.0, 1.0))
    return eng, sd, x


def test_forward_matches_oracle(setup):
    eng, sd, x = setup
    assert list(eng.reference_state_dict().keys()) == [k for k, _ in ro.frcnn_state_keys()]
    out = eng.forward(x.to(dev()), training=False)
    torch.cuda.synchronize()
    p = eng._last_plan
    with torch.no_grad():
        ref = ro.frcnn_forward(sd, x)
    assert [(f.h, f.w) for f in p.features] == [tuple(f.shape[-2:]) for f in ref["features"]]
    for li, (a, r) in enumerate(zip(p.features, ref["features"])):
        assert rel(nchw(a), r) < 4e-2, ("P", li + 2, rel(nchw(a), r))
    assert out["cls_logits"].shape == ref["objectness"].shape and out["bbox_regression"].shape == ref["deltas"].shape
    assert rel(out["cls_logits"].cpu(), ref["objectness"]) < 4e-2
    assert rel(out["bbox_regression"].cpu(), ref["deltas"]) < 4e-2
    # NCHW fp32 views handed to the RoI heads
    for f, r in zip(eng.feature_maps_nchw(4), ref["features"]):
        assert rel(f.cpu(), r) < 4e-2


def test_backward_wiring_with_roi_gradients(setup):
    """Cotangents on the RPN outputs AND on P2..P5 (the RoIAlign return path): parameter gradients vs fp32 autograd of the oracle."""
    eng, sd, x = setup
    sdg = {k: v.clone() for k, v in sd.items()}
    train = [s for s in eng.specs if s.trainable]
    for s in train:
        sdg[s.name + ".weight"].requires_grad_(True)
        if s.bias:
            sdg[s.name + ".bias"].requires_grad_(True)
    ref = ro.frcnn_forward(sdg, x)
    c1 = torch.from_numpy(detrand.uniform(11, tuple(ref["objectness"].shape), -1.0, 1.0)) * 1e-2
    c2 = torch.from_numpy(detrand.uniform(12, tuple(ref["deltas"].shape), -1.0, 1.0)) * 1e-2
    cf = [torch.from_numpy(detrand.uniform(20 + i, tuple(f.shape), -1.0, 1.0)) * 1e-3 for i, f in enumerate(ref["features"][:4])]
    loss = (ref["objectness"] * c1).sum() + (ref["deltas"] * c2).sum() + sum((f * c).sum() for f, c in zip(ref["features"][:4], cf))
    loss.backward()
    eng.forward(x.to(dev()), training=True)
    eng.backward(c1.to(dev()), c2.to(dev()), [c.to(dev()) for c in cf])
    torch.cuda.synchronize()
    got = eng.reference_state_dict(grads=True)
    for s in train:
        for suffix in ([".weight", ".bias"] if s.bias else [".weight"]):
            k = s.name + suffix
            g, r = got[k].cpu(), sdg[k].grad
            c, ratio = cos(g, r), float(g.double().norm() / (r.double().norm() + 1e-30))
            assert c > 0.94 and 0.9 < ratio < 1.1, (k, c, ratio)
    assert cos(got["rpn.head.cls_logits.weight"].cpu(), sdg["rpn.head.cls_logits.weight"].grad) > 0.999
    # the RoI gradient path alone: zero RPN cotangents, gradient only through P2..P5
    eng.forward(x.to(dev()), training=True)
    eng.backward(torch.zeros_like(c1).to(dev()), torch.zeros_like(c2).to(dev()), [c.to(dev()) for c in cf])
    torch.cuda.synchronize()
    g2 = eng.reference_state_dict(grads=True)
    assert float(g2["rpn.head.conv.weight"].abs().max()) == 0.0
    assert float(g2["backbone.fpn.layer_blocks.0.weight"].abs().max()) > 0.0 and float(g2["backbone.body.layer2.0.conv1.weight"].abs().max()) > 0.0


def test_fasterrcnn_module_train_and_eval(setup):
    from object_detectors_amd.tvision.frcnn import fasterrcnn_resnet50_fpn
    _eng, sd, x = setup
    torch.manual_seed(0)
    m = fasterrcnn_resnet50_fpn(num_classes=91, device=dev(), rpn_pre_nms_top_n_train=300, rpn_post_nms_top_n_train=200,
                                rpn_pre_nms_top_n_test=300, rpn_post_nms_top_n_test=100, box_batch_size_per_image=64)
    full = dict(sd)
    for k, v in m.state_dict().items():
        if k.startswith("roi_heads."):
            full[k] = v
    m.load_state_dict(full)
    assert set(k for k in m.state_dict() if k.startswith("roi_heads.")) == {
        "roi_heads.box_head.fc6.weight", "roi_heads.box_head.fc6.bias", "roi_heads.box_head.fc7.weight", "roi_heads.box_head.fc7.bias",
        "roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.cls_score.bias", "roi_heads.box_predictor.bbox_pred.weight",
        "roi_heads.box_predictor.bbox_pred.bias"}
    t = [{"boxes": torch.tensor([[10.0, 12.0, 70.0, 90.0], [40.0, 30.0, 120.0, 100.0]], device=dev()), "labels": torch.tensor([5, 17], device=dev())}
         for _ in range(BS)]
    m.train()
    losses = m(x.to(dev()), t)
    torch.cuda.synchronize()
    assert set(losses) == {"loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg"}
    assert all(bool(torch.isfinite(v)) for v in losses.values())
    assert 0.0 < float(losses["loss_classifier"]) < 50.0 and float(losses["loss_objectness"]) > 0.0
    assert float(m.engine.flat_g.abs().sum()) > 0 and all(p.grad is not None for p in m.head_parameters())
    with pytest.raises(ValueError):
        m(x.to(dev()))
    m.eval()
    det = m([xi.to(dev()) for xi in x])
    assert len(det) == BS and all(set(d) == {"boxes", "labels", "scores"} for d in det)
    for d in det:
        assert d["boxes"].shape[0] == d["scores"].shape[0] == d["labels"].shape[0] <= 100
        assert d["boxes"].shape[0] == 0 or (bool((d["labels"] >= 1).all()) and bool((d["scores"][:-1] >= d["scores"][1:]).all()))
