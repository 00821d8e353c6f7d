"""oracle/retina_oracle.py (torch fp32 restatement of ResNet-50 body + RetinaNetHead) against tests/golden/g12_retinanet.npz,
which holds outputs of the reference's own modules (utilities/resnet.py ResNet, tvision/retinanet.py RetinaNetHead) on
deterministic weights (tools/make_golden.py:g12_retinanet)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import detrand  # noqa: E402
from oracle import retina_oracle as ro  # noqa: E402


def test_state_keys_layout():
    keys = ro.state_keys()
    names = [k for k, _ in keys]
    assert len(names) == len(set(names))
    assert names[0] == "backbone.body.conv1.weight" and names[-1] == "head.regression_head.bbox_reg.bias"
    nparam = sum(int(np.prod(s)) for k, s in keys if not k.endswith(("running_mean", "running_var")))
    # ResNet-50 body without fc (23.5 M) + FPN/extra blocks + RetinaNet heads (91 classes, 9 anchors)
    assert 33_000_000 < nparam < 35_000_000


def test_body_matches_reference(golden):
    g = golden("g12_retinanet")
    seed, xseed, px = (int(v) for v in g["meta"])
    sd = ro.det_state(seed)
    x = torch.from_numpy(detrand.uniform(xseed, (2, 3, px, px), -2.0, 2.0))
    with torch.no_grad():
        feats = ro.body_forward(sd, x)
    for li, f in enumerate(feats, 2):
        assert tuple(f.shape) == tuple(g[f"c{li}_shape"])
        np.testing.assert_allclose(ro.sample(f), g[f"c{li}_sample"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(float(f.double().norm()), float(g[f"c{li}_norm"]), rtol=1e-5)


def test_body_gradients_match_reference(golden):
    g = golden("g12_retinanet")
    seed = int(g["meta"][0])
    sd = ro.det_state(seed)
    names = ["layer2.0.conv1.weight", "layer2.0.downsample.0.weight", "layer3.5.conv2.weight", "layer4.2.conv3.weight"]
    for n in names:
        sd["backbone.body." + n].requires_grad_(True)
    xg = torch.from_numpy(detrand.uniform(7101, (2, 256, 16, 16), -1.0, 1.0)).requires_grad_(True)
    t = xg
    for li, nb in ((2, 4), (3, 6), (4, 3)):
        for b in range(nb):
            t = ro.bottleneck(t, sd, f"backbone.body.layer{li}.{b}", 2 if b == 0 else 1)
    cot = torch.from_numpy(detrand.uniform(7102, tuple(t.shape), -1.0, 1.0))
    (t * cot).sum().backward()
    np.testing.assert_allclose(ro.sample(xg.grad), g["l2in_grad_sample"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(float(xg.grad.double().norm()), float(g["l2in_grad_norm"]), rtol=1e-4)
    for n in names:
        gr = sd["backbone.body." + n].grad
        np.testing.assert_allclose(ro.sample(gr), g["grad_" + n + "_sample"], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(float(gr.double().norm()), float(g["grad_" + n + "_norm"]), rtol=1e-4)


def test_head_matches_reference(golden):
    g = golden("g12_retinanet")
    seed = int(g["meta"][0])
    sd = ro.det_state(seed)
    feats = [torch.from_numpy(detrand.uniform(7200 + l, (2, 256, hw, hw), -1.0, 1.0)) for l, hw in enumerate((8, 4, 2, 1, 1))]
    with torch.no_grad():
        cl, br = ro.head_forward(sd, feats)
    for k, t in (("cls_logits", cl), ("bbox_regression", br)):
        assert tuple(t.shape) == tuple(g[f"head_{k}_shape"])
        np.testing.assert_allclose(ro.sample(t, 256), g[f"head_{k}_sample"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(float(t.double().norm()), float(g[f"head_{k}_norm"]), rtol=1e-5)


def test_full_forward_shapes():
    sd = ro.det_state(1)
    with torch.no_grad():
        out = ro.forward(sd, torch.rand(1, 3, 64, 64))
    assert [tuple(f.shape[-2:]) for f in out["features"]] == [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    assert out["cls_logits"].shape == (1, (64 + 16 + 4 + 1 + 1) * 9, 91) and out["bbox_regression"].shape[-1] == 4
