"""Faster R-CNN training-side mirrors (tvision/rpn.py, tvision/roi_heads.py) on the GPU against the reference fixture g13 and the
oracle: bit-exact labels / matched indices, losses within 1e-4."""
import numpy as np
import pytest

from oracle import tv_oracle as tv

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


class FirstK:
    """deterministic stand-in for the RNG sampler (same rule as tools/make_golden.py:g13_frcnn)."""

    def __call__(self, lbls):
        pos, neg = [], []
        for l in lbls:
            p, n = torch.zeros_like(l, dtype=torch.uint8), torch.zeros_like(l, dtype=torch.uint8)
            p[torch.where(l >= 1)[0][:8]] = 1
            n[torch.where(l == 0)[0][:24]] = 1
            pos.append(p)
            neg.append(n)
        return pos, neg


def test_rpn_targets_and_loss(golden):
    from object_detectors_amd.tvision.rpn import RPNTargets
    g = golden("g13_frcnn")
    rt = RPNTargets()
    anchors = [T(g["anchors"])] * 3
    targets = [{"boxes": T(g[f"gt{i}"])} for i in range(3)]
    labels, mgt = rt.assign_targets_to_anchors(anchors, targets)
    for i in range(3):
        assert np.array_equal(labels[i].cpu().numpy(), g[f"rpn_labels{i}"])
        assert np.array_equal(mgt[i].cpu().numpy(), g[f"rpn_mgt{i}"])
    rt.fg_bg_sampler = FirstK()
    reg = rt.box_coder.encode(mgt, anchors)
    lo, lb = rt.compute_loss(T(g["rpn_obj"]), T(g["rpn_deltas"]), labels, reg)
    np.testing.assert_allclose([float(lo), float(lb)], g["rpn_losses"], rtol=1e-4)
    out = rt.losses(T(g["rpn_obj"]), T(g["rpn_deltas"]), anchors, targets)
    np.testing.assert_allclose([float(out["loss_objectness"]), float(out["loss_rpn_box_reg"])], g["rpn_losses"], rtol=1e-4)


def test_roi_targets_and_fastrcnn_loss(golden):
    from object_detectors_amd.tvision.roi_heads import RoIHeadTargets, fastrcnn_loss
    g = golden("g13_frcnn")
    rh = RoIHeadTargets()
    props = [T(np.concatenate([g["anchors"][:200], g[f"gt{i}"]])) for i in range(2)]
    mi, lab = rh.assign_targets_to_proposals(props, [T(g[f"gt{i}"]) for i in range(2)], [T(g[f"roi_gl{i}"]) for i in range(2)])
    for i in range(2):
        assert np.array_equal(mi[i].cpu().numpy(), g[f"roi_mi{i}"]) and np.array_equal(lab[i].cpu().numpy(), g[f"roi_lab{i}"])
    c, b = fastrcnn_loss(T(g["frcnn_logits"]), T(g["frcnn_breg"]), [T(g["frcnn_labels"])], [T(g["frcnn_tgt"])], loss_type="ce")
    np.testing.assert_allclose([float(c), float(b)], g["frcnn_losses_ce"], rtol=1e-4)
    for lt in ("bce", "focal_loss"):
        c, b = fastrcnn_loss(T(g["frcnn_logits"]), T(g["frcnn_breg"]), [T(g["frcnn_labels"])], [T(g["frcnn_tgt"])], loss_type=lt)
        oc, ob = tv.fastrcnn_loss(g["frcnn_logits"], g["frcnn_breg"], g["frcnn_labels"], g["frcnn_tgt"], lt)
        np.testing.assert_allclose([float(c), float(b)], [oc, ob], rtol=1e-4)
    with pytest.raises(NotImplementedError):
        fastrcnn_loss(T(g["frcnn_logits"]), T(g["frcnn_breg"]), [T(g["frcnn_labels"])], [T(g["frcnn_tgt"])], loss_type="gombit")


def test_select_training_samples_shapes(golden):
    """select_training_samples with the real RNG sampler: structural invariants (rois_heads.py:688-713)."""
    from object_detectors_amd.tvision.roi_heads import RoIHeadTargets
    g = golden("g13_frcnn")
    rh = RoIHeadTargets(batch_size_per_image=64, positive_fraction=0.25)
    props = [T(g["anchors"][:300]) for _ in range(2)]
    targets = [{"boxes": T(g[f"gt{i}"]), "labels": T(g[f"roi_gl{i}"])} for i in range(2)]
    torch.manual_seed(0)
    p, mi, lab, reg = rh.select_training_samples(props, targets)
    for i in range(2):
        n = p[i].shape[0]
        assert n <= 64 and lab[i].shape[0] == n and mi[i].shape[0] == n and reg[i].shape == (n, 4)
        assert int((lab[i] > 0).sum()) <= 16 and int((lab[i] >= 0).all()) == 1
        gtb = targets[i]["boxes"][mi[i]]
        want = tv.encode_boxes(gtb.cpu().numpy(), p[i].cpu().numpy(), (10, 10, 5, 5))
        np.testing.assert_allclose(reg[i].cpu().numpy(), want, rtol=1e-5, atol=1e-6)
        # every GT box was appended to the proposals, so each is its own perfect match and labelled with its class
        assert int((lab[i] > 0).sum()) >= 1
