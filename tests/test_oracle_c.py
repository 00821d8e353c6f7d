"""The C / OpenMP form of the CPU oracle (oracle/c/box_oracle.c) against the numpy oracle and the reference's own outputs
(tests/golden/g5_7_tvision.npz: Matcher and BoxCoder results produced by the imported reference, tools/make_golden.py)."""
import numpy as np
import pytest

from oracle import box_oracle_c as bc
from oracle import detrand
from oracle import tv_oracle as tv
from tests.test_oracle_tv import build_anchors


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not bc.available():
        bc.build()
    assert bc.available()


def _boxes(seed, n, span=400.0):
    ctr = detrand.uniform(seed, (n, 2), 0, span)
    wh = detrand.uniform(seed + 1, (n, 2), 2, 80)
    return np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)


@pytest.mark.parametrize("tag,anc", [("retina", "retina800"), ("rpn", "frcnn800"), ("roi", None), ("retina_m1", "retina800"), ("retina_m20", "retina800")])
def test_matcher_against_the_reference_fixture(golden, tag, anc):
    g = golden("g5_7_tvision")
    anchors = g["match_roi_anchors"] if anc is None else build_anchors(g, anc)
    hi, lo, lowq = g[f"match_{tag}_cfg"]
    q = bc.box_iou(g[f"match_{tag}_gt"], anchors)
    assert np.array_equal(q, tv.box_iou(g[f"match_{tag}_gt"], anchors))                      # same float32 operations: bit-identical
    m = bc.matcher(q, hi, lo, bool(lowq))
    nz = np.nonzero(m != -1)[0]
    assert np.array_equal(nz, g[f"match_{tag}_nz_idx"]) and np.array_equal(m[nz], g[f"match_{tag}_nz_val"])
    assert np.array_equal(m, tv.matcher(q, hi, lo, bool(lowq)))


def test_matcher_tiny_and_errors(golden):
    g = golden("g5_7_tvision")
    for lowq in (0, 1):
        assert np.array_equal(bc.matcher(g["match_tiny_q"], 0.5, 0.4, bool(lowq)), g[f"match_tiny_out_lowq{lowq}"])
    with pytest.raises(ValueError):
        bc.matcher(np.zeros((0, 5), np.float32), 0.5, 0.4, True)


@pytest.mark.parametrize("tag", ["w1", "w10"])
def test_box_coder_against_the_reference_fixture(golden, tag):
    g = golden("g5_7_tvision")
    w = g[f"coder_{tag}_w"]
    np.testing.assert_allclose(bc.encode_boxes(g[f"coder_{tag}_ref"], g[f"coder_{tag}_prop"], w), g[f"coder_{tag}_enc"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bc.decode_boxes(g[f"coder_{tag}_codes"], g[f"coder_{tag}_prop"], w), g[f"coder_{tag}_dec"], rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("n", [1, 57, 1200])
def test_nms_and_batched_nms_equal_the_numpy_oracle(n):
    boxes, scores = _boxes(3 * n, n), detrand.uniform(7 * n, (n,), 0, 1)
    scores[::5] = scores[0]                                                       # ties: lower index first
    idxs = detrand.randint(11 * n, (n,), 0, 5)
    for thr in (0.3, 0.5, 0.7):
        assert np.array_equal(bc.nms(boxes, scores, thr), tv.nms(boxes, scores, thr))
        assert np.array_equal(bc.batched_nms(boxes, scores, idxs, thr), tv.batched_nms(boxes, scores, idxs, thr))
    assert bc.nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).shape == (0,)


def test_sigmoid_focal_loss_sum_equals_the_numpy_oracle():
    x = detrand.uniform(5, (300, 91), -6, 3)
    t = (detrand.uniform(6, (300, 91), 0, 1) > 0.97).astype(np.float32)
    for alpha, gamma in ((0.25, 2.0), (-1.0, 2.0), (0.5, 1.5)):
        loss, grad = tv.sigmoid_focal_loss(x, t, alpha, gamma)
        total, g = bc.sigmoid_focal_loss_sum(x, t, alpha, gamma)
        np.testing.assert_allclose(total, loss.astype(np.float64).sum(), rtol=1e-6)
        np.testing.assert_allclose(g.reshape(x.shape), grad, rtol=1e-6, atol=1e-9)
