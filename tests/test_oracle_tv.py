"""Pin oracle/tv_oracle.py to the reference outputs in tests/golden/g5_7_tvision.npz."""
import numpy as np
import pytest

from oracle import tv_oracle as tv

ANCHOR_TAGS = ["retina800", "frcnn800", "retina800x1216", "retina_small"]


def build_anchors(g, tag):
    sizes = g[f"anc_{tag}_sizes"].tolist()
    ars = g[f"anc_{tag}_ars"].tolist()
    img = tuple(int(v) for v in g[f"anc_{tag}_img"])
    grids = [tuple(int(v) for v in r) for r in g[f"anc_{tag}_grids"]]
    return tv.anchors(sizes, ars, img, grids)


@pytest.mark.parametrize("tag", ANCHOR_TAGS)
def test_anchor_generator(golden, tag):
    g = golden("g5_7_tvision")
    a = build_anchors(g, tag)
    assert a.shape[0] == int(g[f"anc_{tag}_n"][0])
    assert np.array_equal(a[:64], g[f"anc_{tag}_head"])
    assert np.array_equal(a[-64:], g[f"anc_{tag}_tail"])
    assert np.array_equal(a[::1009], g[f"anc_{tag}_sample"])
    np.testing.assert_allclose(a.astype(np.float64).sum(0), g[f"anc_{tag}_sum"])
    if f"anc_{tag}_all" in g.files:
        assert np.array_equal(a, g[f"anc_{tag}_all"])


def test_anchor_counts(golden):
    g = golden("g5_7_tvision")
    assert build_anchors(g, "retina800").shape[0] == 120087
    assert build_anchors(g, "frcnn800").shape[0] == 159882


@pytest.mark.parametrize("tag,anc", [("retina", "retina800"), ("rpn", "frcnn800"), ("roi", None),
                                     ("retina_m1", "retina800"), ("retina_m20", "retina800")])
def test_matcher(golden, tag, anc):
    g = golden("g5_7_tvision")
    anchors = g["match_roi_anchors"] if anc is None else build_anchors(g, anc)
    hi, lo, lowq = g[f"match_{tag}_cfg"]
    m = tv.matcher(tv.box_iou(g[f"match_{tag}_gt"], anchors), hi, lo, bool(lowq))
    assert m.shape[0] == int(g[f"match_{tag}_n"][0])
    nz = np.nonzero(m != -1)[0]
    assert np.array_equal(nz, g[f"match_{tag}_nz_idx"])
    assert np.array_equal(m[nz], g[f"match_{tag}_nz_val"])


def test_matcher_tiny_and_errors(golden):
    g = golden("g5_7_tvision")
    for lowq in (0, 1):
        assert np.array_equal(tv.matcher(g["match_tiny_q"], 0.5, 0.4, bool(lowq)), g[f"match_tiny_out_lowq{lowq}"])
    with pytest.raises(ValueError):
        tv.matcher(np.zeros((0, 5), np.float32), 0.5, 0.4, True)
    with pytest.raises(ValueError):
        tv.matcher(np.zeros((3, 0), np.float32), 0.5, 0.4, True)


@pytest.mark.parametrize("tag", ["w1", "w10"])
def test_box_coder(golden, tag):
    g = golden("g5_7_tvision")
    w = g[f"coder_{tag}_w"]
    np.testing.assert_allclose(tv.encode_boxes(g[f"coder_{tag}_ref"], g[f"coder_{tag}_prop"], w),
                               g[f"coder_{tag}_enc"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tv.decode_boxes(g[f"coder_{tag}_codes"], g[f"coder_{tag}_prop"], w),
                               g[f"coder_{tag}_dec"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(tv.decode_boxes(g[f"coder_{tag}_codes3"], g[f"coder_{tag}_prop"], w),
                               g[f"coder_{tag}_dec3"], rtol=1e-5, atol=1e-3)


def test_nms_properties():
    from oracle import detrand
    c = detrand.uniform(5, (500, 2), 0, 300)
    s = detrand.uniform(6, (500, 2), 5, 120)
    boxes = np.concatenate([c, c + s], 1)
    scores = detrand.uniform(7, (500,), 0, 1)
    keep = tv.nms(boxes, scores, 0.5)
    assert (np.diff(scores[keep]) <= 0).all()
    iou = tv.box_iou(boxes[keep], boxes[keep])
    np.fill_diagonal(iou, 0)
    assert (iou <= 0.5).all()
    # every suppressed box overlaps a kept, higher-scored box
    sup = np.setdiff1d(np.arange(500), keep)
    q = tv.box_iou(boxes[sup], boxes[keep])
    assert ((q > 0.5) & (scores[keep][None, :] >= scores[sup][:, None])).any(1).all()
    # idempotence
    assert np.array_equal(tv.nms(boxes[keep], scores[keep], 0.5), np.arange(len(keep)))
    # batched: classes never suppress each other
    idxs = np.arange(500) % 7
    kb = tv.batched_nms(boxes, scores, idxs, 0.5)
    for cls in range(7):
        sel = np.nonzero(idxs == cls)[0]
        assert np.array_equal(np.sort(sel[tv.nms(boxes[sel], scores[sel], 0.5)]), np.sort(kb[idxs[kb] == cls]))
