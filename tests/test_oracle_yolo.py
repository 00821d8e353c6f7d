"""Pin the CPU oracle (oracle/yolo_oracle.py) to the reference's own outputs (tests/golden)."""
import numpy as np
import pytest

from oracle import yolo_oracle as yo
from tests.helpers import YOLO_CASES, YOLO_FULL, yolo_case


def test_bbox_iou_matches_reference(golden):
    g = golden("g1_bbox_iou")
    for t in range(4):
        got = yo.bbox_iou(g["bb1"], g["bb2"], t)
        np.testing.assert_allclose(got, g[f"iou_type{t}"], rtol=1e-6, atol=1e-7, equal_nan=True)
        got = yo.bbox_iou(g["e1"], g["e2"], t)
        np.testing.assert_allclose(got, g[f"elem_type{t}"], rtol=1e-6, atol=1e-7)
    # IoU / GIoU are pure + - * / max min : bit-exact
    for t in (0, 1):
        assert np.array_equal(yo.bbox_iou(g["bb1"], g["bb2"], t), g[f"iou_type{t}"], equal_nan=True)
    assert np.array_equal(yo.bbox_iou(g["xyxy_a"], g["xyxy_b"], 0, xcycwh=False), g["xyxy_iou0"])


def test_nms_majority_matches_reference(golden):
    g = golden("g2_nms_majority")
    names = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    assert len(names) >= 10
    for n in names:
        for tag, thr in (("", 0.6), ("_t45", 0.45)):
            key = n + "_out" + tag
            if key not in g.files:
                continue
            got, _ = yo.nms_majority(g[n + "_in"], thr)
            assert np.array_equal(got, g[key]), n + tag   # keep set, order and relabels bit-exact


def test_focal_loss_matches_reference(golden):
    g = golden("g9_focal")
    for gamma, alpha in ((1.0, 0.5), (1.5, 0.25), (2.0, 0.25)):
        loss, grad = yo.focal_loss(g["x"], g["t"], gamma, alpha)
        np.testing.assert_allclose(loss, g[f"g{gamma}_a{alpha}_None"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(loss.sum(), g[f"g{gamma}_a{alpha}_sum"], rtol=2e-5)
        np.testing.assert_allclose(grad, g[f"g{gamma}_a{alpha}_None_grad"], rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("name", YOLO_CASES)
def test_yolo_loss_matches_reference(golden, name):
    g = golden("g3_yolo_forw")
    spec, heads, targets = yolo_case(g, name)
    r = yo.yolo_loss(spec, heads, targets, want_grad=name in YOLO_FULL)
    # indices: bit-exact
    assert np.array_equal(np.concatenate(r["obj_idx"]), g[name + "_obj_idx"])
    bits = np.packbits(r["noobj"].astype(np.uint8), axis=1, bitorder="little")
    assert np.array_equal(bits, g[name + "_noobj_bits"])
    np.testing.assert_allclose(r["tgt"], g[name + "_tgt"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["loss"], g[name + "_loss"], rtol=1e-4)
    np.testing.assert_allclose(r["sub_losses"], g[name + "_sub_losses"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r["stats"], g[name + "_stats"], rtol=1e-4, atol=1e-6)
    if name in YOLO_FULL:
        for k, gr in enumerate(r["grads"]):
            np.testing.assert_allclose(gr, g[f"{name}_grad{k}"], rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("name", YOLO_CASES)
def test_yolo_decode_matches_reference(golden, name):
    g = golden("g3_yolo_forw")
    spec, heads, _ = yolo_case(g, name)
    dec = yo.decode(spec, heads)
    if name in YOLO_FULL:
        np.testing.assert_allclose(dec, g[name + "_decode"], rtol=1e-4, atol=1e-6)
    else:
        flat = dec.reshape(-1)
        np.testing.assert_allclose(flat[::997], g[name + "_decode_sample"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(flat.astype(np.float64).sum(), g[name + "_decode_digest"][0], rtol=1e-5)


def test_postprocess_matches_reference(golden):
    g = golden("g11_postproc")
    g3 = golden("g3_yolo_forw")
    from tests.helpers import synth_heads
    from oracle.yolo_oracle import YoloSpec
    seed, C, img, bs = [int(v) for v in g["meta"]]
    spec = YoloSpec(g3["coco128_anchors"].tolist(), C, img)
    heads = synth_heads(seed, bs, 3, C, (4, 8, 16))
    res = yo.postprocess(yo.decode(spec, heads), float(g["conf"][0]), 0.6)
    assert len(res) == 2
    for e, (cand, fin) in enumerate(res):
        assert cand.shape == g[f"cand{e}"].shape
        np.testing.assert_allclose(cand, g[f"cand{e}"], rtol=1e-4, atol=1e-5)
        # NMS on the reference's own candidates: bit-exact keep set
        fin2, _ = yo.nms_majority(g[f"cand{e}"], 0.6)
        assert np.array_equal(fin2, g[f"final{e}"])
