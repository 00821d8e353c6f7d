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
