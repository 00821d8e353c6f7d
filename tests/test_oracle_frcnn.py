"""oracle/tv_oracle.py Faster R-CNN target / loss restatements against tests/golden/g13_frcnn.npz (outputs of the reference's
RegionProposalNetwork.assign_targets_to_anchors / compute_loss, RoIHeads.assign_targets_to_proposals, fastrcnn_loss)."""
import numpy as np

from oracle import tv_oracle as tv


def _sampled(labels_list):
    pos, neg, off = [], [], 0
    for l in labels_list:
        pos.append(np.nonzero(l >= 1)[0][:8] + off)
        neg.append(np.nonzero(l == 0)[0][:24] + off)
        off += len(l)
    return np.concatenate(pos), np.concatenate(neg)


def test_rpn_assign_and_loss(golden):
    g = golden("g13_frcnn")
    anchors = g["anchors"]
    labels, mgts = [], []
    for i in range(3):
        lab, mgt = tv.rpn_assign(anchors, g[f"gt{i}"])
        assert np.array_equal(lab, g[f"rpn_labels{i}"]) and np.array_equal(mgt, g[f"rpn_mgt{i}"])
        labels.append(lab)
        mgts.append(mgt)
    reg = np.concatenate([tv.encode_boxes(m, anchors, (1, 1, 1, 1)) for m in mgts])
    pos, neg = _sampled(labels)
    lo, lb = tv.rpn_loss(g["rpn_obj"], g["rpn_deltas"], np.concatenate(labels), reg, pos, neg)
    np.testing.assert_allclose([lo, lb], g["rpn_losses"], rtol=2e-5)


def test_roi_assign_and_fastrcnn_loss(golden):
    g = golden("g13_frcnn")
    for i in range(2):
        props = np.concatenate([g["anchors"][:200], g[f"gt{i}"]])
        mi, lab = tv.roi_assign(props, g[f"gt{i}"], g[f"roi_gl{i}"])
        assert np.array_equal(mi, g[f"roi_mi{i}"]) and np.array_equal(lab, g[f"roi_lab{i}"])
    c, b = tv.fastrcnn_loss(g["frcnn_logits"], g["frcnn_breg"], g["frcnn_labels"], g["frcnn_tgt"], "ce")
    np.testing.assert_allclose([c, b], g["frcnn_losses_ce"], rtol=2e-5)


LOSS_CASES = [("ce", 1.0), ("bce", 1.0), ("focal_loss", 1.0), ("gombit", 1.0), ("gombit_fl", 1.0), ("gombit", 3.0), ("gombit", -1.0)]


def loss_case_inputs(g, lt, scale):
    """logits / tag of one fixture case (tools/make_golden.py:g13_frcnn)."""
    logits = g["frcnn_logits"] * np.float32(scale) if scale > 0 else g["frcnn_logits"] * np.float32(0.5) - np.float32(5.0)
    tag = lt + ("_x3" if scale == 3.0 else "_lo" if scale < 0 else "")
    return logits.astype(np.float32), tag


def test_fastrcnn_loss_variants_with_tfidf_and_weights(golden):
    """Every loss_type of the reference's fastrcnn_loss as RoIHeads.forward calls it (tf-idf row on the logits, class weights for 'ce'),
    losses and logit gradients incl. both sides of the gombit "/4 above 5" branch."""
    g = golden("g13_frcnn")
    tfidf, cw = g["frcnn_tfidf"], g["frcnn_cw"]
    for lt, scale in LOSS_CASES:
        logits, tag = loss_case_inputs(g, lt, scale)
        c, b, gl, gb = tv.fastrcnn_loss(tfidf * logits, g["frcnn_breg"], g["frcnn_labels"], g["frcnn_tgt"], lt, weights=cw if lt == "ce" else None,
                                        want_grad=True)
        np.testing.assert_allclose([c, b], g[f"frcnn_w_losses_{tag}"], rtol=3e-5, err_msg=tag)
        ref = g[f"frcnn_w_glogits_{tag}"]
        np.testing.assert_allclose(gl * tfidf, ref, rtol=2e-4, atol=2e-7 * float(np.abs(ref).max()) + 1e-9, err_msg=tag)   # chain rule through tfidf * logits
        if lt == "ce":
            np.testing.assert_allclose(gb, g["frcnn_w_gbreg"], rtol=1e-5, atol=1e-9)
    assert g["frcnn_w_losses_gombit_lo"][0] < 5 < g["frcnn_w_losses_gombit"][0] * 4


def test_minibatch_tfidf_and_postprocess_detections(golden):
    g = golden("g13_frcnn")
    labels = [g["pp_labels0"], g["pp_labels1"]]
    for norm in (0, 2):
        np.testing.assert_allclose(tv.minibatch_tfidf(labels, 21, norm), g[f"pp_minibatch_tfidf_norm{norm}"], rtol=1e-6)
    props = [g["pp_props0"], g["pp_props1"]]
    for lt in ("ce", "bce", "gombit"):
        res = tv.roi_postprocess_detections(g["pp_logits"], g["pp_breg"], props, [(512, 640), (480, 512)], g["pp_tfidf_post"], lt, 0.05, 0.5, 20)
        for i, (b, s, l) in enumerate(res):
            assert np.array_equal(l, g[f"pp_{lt}_labels{i}"]), (lt, i)
            np.testing.assert_allclose(s, g[f"pp_{lt}_scores{i}"], rtol=2e-6)
            np.testing.assert_allclose(b, g[f"pp_{lt}_boxes{i}"], rtol=1e-5, atol=1e-4)
            assert len(l) == 20
