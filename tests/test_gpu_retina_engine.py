"""RetinaNet-ResNet50-FPN engine (object_detectors_amd/tvision/engine.py) against oracle/retina_oracle.py (torch fp32, pinned to the
reference's ResNet / RetinaNetHead by tests/golden/g12_retinanet.npz) and oracle/tv_oracle.py (loss)."""
import numpy as np
import pytest

from oracle import detrand
from oracle import retina_oracle as ro
from oracle import tv_oracle as tv

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

PX, BS, SEED = 128, 2, 7000


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
    from object_detectors_amd.tvision.engine import RetinaNetEngine
    sd = ro.det_state(SEED)
    eng = RetinaNetEngine(91, 9, 3, device=dev(), seed=0)
    eng.load_reference_state_dict(sd)
    x = torch.from_numpy(detrand.uniform(4242, (BS, 3, PX, PX), 0.0, 1.0))
    return eng, sd, x


def test_state_dict_roundtrip(setup):
    eng, sd, _x = setup
    out = eng.reference_state_dict()
    assert list(out.keys()) == [k for k, _ in ro.state_keys()]
    for k, v in sd.items():
        assert torch.equal(out[k].cpu(), v), k


def test_forward_matches_oracle(setup):
    eng, sd, x = setup
    out = eng.forward(x.to(dev()), training=False)
    torch.cuda.synchronize()
    p = eng._last_plan
    with torch.no_grad():
        ref = ro.forward(sd, x)
    for li, (a, r) in enumerate(zip(p.body, ref["body"]), 2):
        assert rel(nchw(a), r) < 4e-2, ("C", li, rel(nchw(a), r))
    for li, (a, r) in enumerate(zip(p.features, ref["features"]), 3):
        assert rel(nchw(a), r) < 4e-2, ("P", li, rel(nchw(a), r))
    assert rel(out["cls_logits"].cpu(), ref["cls_logits"]) < 4e-2
    assert rel(out["bbox_regression"].cpu(), ref["bbox_regression"]) < 4e-2
    assert cos(out["cls_logits"].cpu() - ref["cls_logits"].mean(), ref["cls_logits"] - ref["cls_logits"].mean()) > 0.999


def _grad_report(eng, sdg):
    got = eng.reference_state_dict(grads=True)
    rep = {}
    for s in eng.specs:
        if not s.trainable:
            continue
        for suffix in ([".weight", ".bias"] if s.bias else [".weight"]):
            k = s.name + suffix
            g, r = got[k].cpu(), sdg[k].grad
            rep[k] = (cos(g, r), float(g.double().norm() / (r.double().norm() + 1e-30)))
    assert "backbone.body.layer1.0.conv1.weight" not in got          # frozen (trainable_layers=3)
    return rep


def _oracle_with_grads(eng, sd, x):
    sdg = {k: v.clone() for k, v in sd.items()}
    for s in eng.specs:
        if s.trainable:
            sdg[s.name + ".weight"].requires_grad_(True)
            if s.bias:
                sdg[s.name + ".bias"].requires_grad_(True)
    return sdg, ro.forward(sdg, x)


def test_backward_wiring_random_cotangents(setup):
    """Random-sign cotangents on every logit: the parameter gradients are sums of cancelling terms, so bf16 rounding shows up
    as noise (measured: cos 0.96-0.99 decaying smoothly with depth, 0.99995 at the last conv); a wiring error (missing
    branch, wrong accumulation) would break a whole sub-tree instead."""
    eng, sd, x = setup
    sdg, ref = _oracle_with_grads(eng, sd, x)
    c1 = torch.from_numpy(detrand.uniform(11, tuple(ref["cls_logits"].shape), -1.0, 1.0)) * 1e-2
    c2 = torch.from_numpy(detrand.uniform(12, tuple(ref["bbox_regression"].shape), -1.0, 1.0)) * 1e-2
    ((ref["cls_logits"] * c1).sum() + (ref["bbox_regression"] * c2).sum()).backward()
    eng.forward(x.to(dev()), training=True)
    eng.backward(c1.to(dev()), c2.to(dev()))
    torch.cuda.synchronize()
    rep = _grad_report(eng, sdg)
    for k, (c, ratio) in rep.items():
        assert c > 0.94 and 0.9 < ratio < 1.1, (k, c, ratio)
    assert rep["head.classification_head.cls_logits.weight"][0] > 0.9995 and rep["head.regression_head.bbox_reg.weight"][0] > 0.9995


def _targets():
    rng = np.random.default_rng(0)
    targets, gts = [], []
    for i in range(BS):
        m = 3 + i
        tl = rng.uniform(0, PX * 0.5, (m, 2)).astype(np.float32)
        wh = rng.uniform(PX * 0.1, PX * 0.45, (m, 2)).astype(np.float32)
        boxes = np.concatenate([tl, tl + wh], 1)
        labels = rng.integers(1, 91, (m,)).astype(np.int64)
        gts.append((boxes, labels))
        targets.append({"boxes": torch.from_numpy(boxes).to(dev()), "labels": torch.from_numpy(labels).to(dev())})
    return targets, gts


def test_train_step_gradients_match_oracle(setup):
    """Whole training step (forward, matcher, focal + L1 loss, backward) vs torch fp32 autograd of the oracle network driven by the
    oracle loss gradient: every trainable parameter's gradient within cos > 0.995 and 3 % in norm
    (measured >= 0.9991 / 0.6 % on the body and heads, 0.997 on the 2x2 / 1x1 P6, P7 maps of this small input)."""
    eng, sd, x = setup
    sdg, ref = _oracle_with_grads(eng, sd, x)
    targets, gts = _targets()
    eng.forward(x.to(dev()), training=True)
    anchors = eng._last_plan.anchors.cpu().numpy()
    cl, rl, _mis, (gc, gr) = tv.retinanet_loss(ref["cls_logits"].detach().numpy(), ref["bbox_regression"].detach().numpy(), anchors, gts)
    ((ref["cls_logits"] * torch.from_numpy(gc)).sum() + (ref["bbox_regression"] * torch.from_numpy(gr)).sum()).backward()
    losses = eng.train_step(x.to(dev()), targets)
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses.cpu().numpy(), [cl, rl], rtol=5e-3)
    for k, (c, ratio) in _grad_report(eng, sdg).items():
        # P6 / P7 are 2x2 / 1x1 maps at this input size: their weight gradients sum over 8 / 2 pixels, so a single ReLU mask that
        # flips between the bf16 engine and the fp32 oracle moves them visibly (measured 0.988-0.9999 depending on the tile
        # configuration's summation order); everything else is >= 0.9988
        lo = 0.97 if ".extra_blocks." in k else 0.995
        assert c > lo and 0.97 < ratio < 1.03, (k, c, ratio)


def test_train_step_losses_and_matching(setup):
    eng, sd, x = setup
    targets, gts = _targets()
    losses = eng.train_step(x.to(dev()), targets)
    torch.cuda.synchronize()
    p = eng._last_plan
    anchors = p.anchors.cpu().numpy()
    cl, rl, mis, (gc, gr) = tv.retinanet_loss(p.logits.cpu().numpy(), p.bbox_reg.cpu().numpy(), anchors, gts)
    assert np.array_equal(p.matched.cpu().numpy(), np.stack(mis))
    np.testing.assert_allclose(losses.cpu().numpy(), [cl, rl], rtol=3e-4)
    np.testing.assert_allclose(p.gbbox.cpu().numpy(), gr, rtol=1e-5, atol=1e-9)
    # the class gradient is written by the loss kernel straight into the bf16 level buffers of the cls_logits backward (no fp32 tensor):
    # every level holds the bf16 rounding of the oracle's gradient rows
    row0 = 0
    for lvl, rows in enumerate(p.level_rows):
        gl = p.head_grads[("cls_logits", lvl)][..., :9 * 91].float().reshape(BS, -1, 91).cpu().numpy()
        np.testing.assert_allclose(gl, gc[:, row0:row0 + rows], rtol=5e-3, atol=1e-7)
        row0 += rows
    assert float(eng.flat_g.abs().sum()) > 0


def test_module_mirror_eval_and_train(setup):
    from object_detectors_amd.tvision.retinanet import retinanet_resnet50_fpn
    _eng, sd, x = setup
    m = retinanet_resnet50_fpn(num_classes=91, device=dev())
    m.load_state_dict(sd)
    m.eval()
    det = m([xi.to(dev()) for xi in x])
    assert len(det) == BS and set(det[0].keys()) == {"boxes", "scores", "labels"}
    for d in det:
        assert d["boxes"].shape[0] == d["scores"].shape[0] == d["labels"].shape[0] <= 300
        if d["scores"].numel() > 1:
            assert bool((d["scores"][:-1] >= d["scores"][1:]).all())
    m.train()
    t = [{"boxes": torch.tensor([[10.0, 12.0, 70.0, 90.0]], device=dev()), "labels": torch.tensor([5], device=dev())} for _ in range(BS)]
    out = m(x.to(dev()), t)
    assert set(out.keys()) == {"classification", "bbox_regression"} and all(torch.isfinite(v) for v in out.values())
    with pytest.raises(ValueError):
        m(x.to(dev()), [{"boxes": torch.tensor([[10.0, 12.0, 5.0, 90.0]], device=dev()), "labels": torch.tensor([5], device=dev())}] * BS)


def test_image_without_ground_truth(setup):
    """retinanet.py:404-407: an image with no boxes is all background (matched = -1): finite losses, zero regression term for it."""
    eng, _sd, x = setup
    t = [{"boxes": torch.zeros((0, 4), device=dev()), "labels": torch.zeros((0,), dtype=torch.int64, device=dev())},
         {"boxes": torch.tensor([[10.0, 12.0, 70.0, 90.0]], device=dev()), "labels": torch.tensor([5], device=dev())}]
    losses = eng.train_step(x.to(dev()), t)
    torch.cuda.synchronize()
    p = eng._last_plan
    assert bool(torch.isfinite(losses).all()) and float(losses[0]) > 0
    assert int((p.matched[0] != -1).sum()) == 0 and int((p.matched[1] >= 0).sum()) > 0
    assert float(eng.last_num_foreground[0]) == 0.0
    assert float(p.gbbox[0].abs().max()) == 0.0 and float(p.gbbox[1].abs().max()) > 0.0
    both_empty = [t[0], t[0]]
    losses = eng.train_step(x.to(dev()), both_empty)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(losses).all()) and float(losses[1]) == 0.0


def test_trainable_layers_5_trains_the_stem_through_the_max_pool():
    """trainable_backbone_layers = 5 (backbone_utils.py:100-104): conv1 (7x7/2) and layer1 receive gradients - through the new max-pool
    backward (gradient to the first maximum of every 3x3 window) and FrozenBN + ReLU of the stem - against fp32 autograd of the oracle."""
    from object_detectors_amd.tvision.engine import RetinaNetEngine
    sd = ro.det_state(SEED)
    eng = RetinaNetEngine(91, 9, 5, device=dev(), seed=0)
    eng.load_reference_state_dict(sd)
    assert all(s.trainable for s in eng.specs)
    x = torch.from_numpy(detrand.uniform(4242, (BS, 3, PX, PX), 0.0, 1.0))
    sdg, ref = _oracle_with_grads(eng, sd, x)
    targets, gts = _targets()
    eng.forward(x.to(dev()), training=True)
    anchors = eng._last_plan.anchors.cpu().numpy()
    cl, rl, _mis, (gc, gr) = tv.retinanet_loss(ref["cls_logits"].detach().numpy(), ref["bbox_regression"].detach().numpy(), anchors, gts)
    ((ref["cls_logits"] * torch.from_numpy(gc)).sum() + (ref["bbox_regression"] * torch.from_numpy(gr)).sum()).backward()
    eng.train_step(x.to(dev()), targets)
    torch.cuda.synchronize()
    got = eng.reference_state_dict(grads=True)
    for k in ("backbone.body.conv1.weight", "backbone.body.layer1.0.conv1.weight", "backbone.body.layer1.0.downsample.0.weight",
              "backbone.body.layer1.2.conv3.weight", "backbone.body.layer2.0.conv2.weight"):
        g, r = got[k].cpu(), sdg[k].grad
        c, ratio = cos(g, r), float(g.double().norm() / (r.double().norm() + 1e-30))
        assert c > 0.99 and 0.95 < ratio < 1.05, (k, c, ratio)
    with pytest.raises(ValueError):
        RetinaNetEngine(91, 9, 6, device=dev(), seed=0)


def test_maxpool_backward_matches_torch():
    import ctypes as C
    import torch.nn.functional as F
    from object_detectors_amd._lib import check, lib
    g = torch.Generator().manual_seed(5)
    for (n, h, w, c) in [(2, 17, 23, 16), (1, 32, 32, 64)]:
        x = torch.randn((n, c, h, w), generator=g).bfloat16().float()
        x[:, :, 4:8, 4:8] = 0.5                                     # ties: the first maximum of a window takes the gradient
        xr = x.clone().requires_grad_(True)
        y = F.max_pool2d(xr, 3, 2, 1)
        gy = torch.randn(tuple(y.shape), generator=g).bfloat16().float()
        y.backward(gy)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev()).bfloat16()
        gd = gy.permute(0, 2, 3, 1).contiguous().to(dev()).bfloat16()
        dx = torch.full((n, h, w, c), 3.0, device=dev(), dtype=torch.bfloat16)
        vp = lambda t: C.c_void_p(t.data_ptr())
        check(lib().mi355det_maxpool3x3s2_bwd(vp(xd), c, vp(gd), c, n, h, w, c, vp(dx), c, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "maxpool_bwd")
        got = dx.float().cpu().permute(0, 3, 1, 2)
        want = xr.grad.bfloat16().float()
        assert float((got - want).abs().max()) <= 1e-2 * float(want.abs().max()), (n, h, w, c)
