"""GPU parity: HIP YOLO criterion / NMS kernels (through the C ABI) vs the CPU oracle and the
reference-generated golden fixtures.  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest

from oracle import detrand
from oracle import yolo_oracle as yo
from tests.helpers import YOLO_CASES, YOLO_FULL, synth_heads, yolo_case

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def make_module(spec):
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    return YOLOForw(anchors=spec.anchors, num_classes=spec.C, img_size=spec.img_size, iou_type=spec.iou_type,
                    idf_logits=spec.idf, class_weights=spec.cw, class_loss=spec.class_loss, reduction=spec.reduction,
                    img_freq=getattr(spec, "img_freq", None)).to(dev())


def to_targets(targets):
    return [{"bbox": torch.from_numpy(b).to(dev()), "category_id": torch.from_numpy(l).to(dev())} for b, l in targets]


def test_bbox_iou_gpu(golden):
    from object_detectors_amd.yolo.utilities import helper
    g = golden("g1_bbox_iou")
    for t in range(4):
        got = helper.bbox_iou(torch.from_numpy(g["bb1"]).to(dev()), torch.from_numpy(g["bb2"]).to(dev()), t).cpu().numpy()
        np.testing.assert_allclose(got, g[f"iou_type{t}"], rtol=1e-5, atol=1e-6, equal_nan=True)
        if t < 2:   # + - * / min max only: bit-exact
            assert np.array_equal(got, g[f"iou_type{t}"], equal_nan=True)
        got = helper.bbox_iou(torch.from_numpy(g["e1"]).to(dev()), torch.from_numpy(g["e2"]).to(dev()), t).cpu().numpy()
        np.testing.assert_allclose(got, g[f"elem_type{t}"], rtol=1e-5, atol=1e-6)
    got = helper.bbox_iou(torch.from_numpy(g["xyxy_a"]).to(dev()), torch.from_numpy(g["xyxy_b"]).to(dev()), 0, xcycwh=False)
    assert np.array_equal(got.cpu().numpy(), g["xyxy_iou0"])


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("name", YOLO_CASES)
def test_yolo_loss_gpu(golden, name, layout):
    g = golden("g3_yolo_forw")
    spec, heads, targets = yolo_case(g, name)
    mod = make_module(spec)
    th = []
    for h in heads:
        t = torch.from_numpy(h).to(dev())
        if layout == "nhwc":   # engine-native memory layout behind an NCHW-shaped view
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        th.append(t.requires_grad_(True))
    loss, sub, stats = mod(th, to_targets(targets))
    loss.backward()
    obj_idx, tgt, noobj, counts = mod.last_assignment
    # indices bit-exact against the REFERENCE fixture (types 0..2; CIoU uses atan)
    if spec.iou_type != 3:
        assert np.array_equal(obj_idx.cpu().numpy(), g[name + "_obj_idx"])
        bits = np.packbits(noobj.cpu().numpy(), axis=1, bitorder="little")
        assert np.array_equal(bits, g[name + "_noobj_bits"])
    np.testing.assert_allclose(tgt.cpu().numpy(), g[name + "_tgt"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss.item(), g[name + "_loss"], rtol=1e-4)
    np.testing.assert_allclose(sub.cpu().numpy(), g[name + "_sub_losses"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(stats.cpu().numpy(), g[name + "_stats"], rtol=1e-4, atol=1e-6)
    for k, t in enumerate(th):
        gr = t.grad.cpu().numpy()
        if name in YOLO_FULL:
            np.testing.assert_allclose(gr, g[f"{name}_grad{k}"], rtol=1e-3, atol=1e-6)
        else:
            flat = gr.reshape(-1)
            np.testing.assert_allclose(flat[::997], g[f"{name}_grad{k}_sample"], rtol=1e-3, atol=1e-6)
            np.testing.assert_allclose(np.abs(flat).astype(np.float64).sum(), g[f"{name}_grad{k}_digest"][1], rtol=1e-4)


@pytest.mark.parametrize("name", YOLO_CASES)
def test_yolo_decode_gpu(golden, name):
    g = golden("g3_yolo_forw")
    spec, heads, _ = yolo_case(g, name)
    mod = make_module(spec)
    with torch.no_grad():
        dec = mod([torch.from_numpy(h).to(dev()) for h in heads]).cpu().numpy()
        dec2 = mod([torch.from_numpy(h).to(dev()).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for h in heads]).cpu().numpy()
    np.testing.assert_allclose(dec2, dec, rtol=1e-5, atol=1e-6)   # NHWC (LDS-tiled) and NCHW (wave-per-row) paths: different sum order
    if name in YOLO_FULL:
        np.testing.assert_allclose(dec, g[name + "_decode"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(dec2, g[name + "_decode"], rtol=1e-4, atol=1e-5)
    else:
        flat = dec.reshape(-1)
        np.testing.assert_allclose(flat[::997], g[name + "_decode_sample"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(flat.astype(np.float64).sum(), g[name + "_decode_digest"][0], rtol=1e-5)


def test_yolo_loss_full_size_vs_oracle():
    """BASELINE config shape (640 px, bs 4 here so the oracle finishes in seconds)."""
    from tests.helpers import synth_targets
    anchors = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
    spec = yo.YoloSpec(anchors, 80, 640)
    heads = synth_heads(31337, 4, 3, 80, (20, 40, 80))
    targets = synth_targets(4242, (7, 1, 20, 7), 80)
    ref = yo.yolo_loss(spec, heads, targets, want_grad=True)
    mod = make_module(spec)
    th = [torch.from_numpy(h).to(dev()).requires_grad_(True) for h in heads]
    loss, sub, stats = mod(th, to_targets(targets))
    loss.backward()
    obj_idx, tgt, noobj, _ = mod.last_assignment
    assert np.array_equal(obj_idx.cpu().numpy(), np.concatenate(ref["obj_idx"]))
    assert np.array_equal(noobj.cpu().numpy().astype(bool), ref["noobj"])
    np.testing.assert_allclose(loss.item(), ref["loss"], rtol=1e-4)
    np.testing.assert_allclose(sub.cpu().numpy(), ref["sub_losses"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(stats.cpu().numpy(), ref["stats"], rtol=1e-4, atol=1e-6)
    for t, gr in zip(th, ref["grads"]):
        np.testing.assert_allclose(t.grad.cpu().numpy(), gr, rtol=2e-3, atol=2e-6)


def test_get_target_empty_image_and_many_gt():
    """ragged batch: an image without GT and one with more GT than one LDS chunk (64)."""
    from tests.helpers import synth_targets
    anchors = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
    spec = yo.YoloSpec(anchors, 80, 416)
    targets = synth_targets(99, (0, 150, 3), 80)
    cx, inw = yo.anchor_table(spec, (13, 26, 52))
    ref_tgt, ref_idx, ref_noobj = yo.get_target(spec, [t for t in targets if len(t[0])], cx, inw)
    mod = make_module(spec)
    tgt, obj, noobj = mod.get_target(to_targets(targets), (13, 26, 52))
    assert np.array_equal(torch.cat(obj).cpu().numpy(), np.concatenate(ref_idx))
    assert noobj[0].all()
    assert np.array_equal(noobj[1:].cpu().numpy(), ref_noobj)
    np.testing.assert_allclose(tgt.cpu().numpy(), ref_tgt, rtol=1e-5, atol=1e-6)


def test_nms_majority_gpu(golden):
    from object_detectors_amd.yolo.utilities import helper
    g = golden("g2_nms_majority")
    names = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    for n in names:
        for tag, thr in (("", 0.6), ("_t45", 0.45)):
            key = n + "_out" + tag
            if key not in g.files:
                continue
            P = torch.from_numpy(g[n + "_in"].copy()).to(dev())
            out = helper.nms_majority(P, thr)
            assert np.array_equal(out.cpu().numpy(), g[key]), n + tag     # keep set, order, relabels: bit-exact
            # in-place relabel of the INPUT, like the reference (helper.py:374-375: the kept row is a view of P): the kept rows of P carry
            # the voted label afterwards, every other row and every other column is untouched
            ref_rows, ref_keep = yo.nms_majority(g[n + "_in"], thr)
            want_in = g[n + "_in"].copy()
            want_in[ref_keep, 5] = ref_rows[:, 5]
            assert np.array_equal(P.cpu().numpy(), want_in), n + tag


def test_nms_majority_large_vs_oracle():
    c = detrand.uniform(71, (6000, 2), 40, 600)
    s = np.exp(detrand.uniform(72, (6000, 2), np.log(8), np.log(250))).astype(np.float32)
    P = np.concatenate([c - s / 2, c + s / 2, detrand.uniform(73, (6000, 1), 0.1, 1), detrand.randint(74, (6000, 1), 0, 80).astype(np.float32)], 1)
    P = P.astype(np.float32)
    from object_detectors_amd.yolo.utilities import helper
    out = helper.nms_majority(torch.from_numpy(P.copy()).to(dev()), 0.6)
    ref, _ = yo.nms_majority(P, 0.6)
    assert np.array_equal(out.cpu().numpy(), ref)
    # idempotence property at full size: NMS of the kept set keeps everything (labels may not change again)
    out2 = helper.nms_majority(out.clone(), 0.6)
    assert out2.shape == out.shape and np.array_equal(out2[:, :5].cpu().numpy(), out[:, :5].cpu().numpy())


def test_nms_majority_equal_scores_radix_path():
    """n >= 4096 takes the radix sort: quantised scores, so the order of equal scores (the reverse of a stable ascending argsort,
    helper.py:308,320) decides which box of a cluster survives."""
    n = 5000
    c = detrand.uniform(81, (n, 2), 40, 1500)
    s = np.exp(detrand.uniform(82, (n, 2), np.log(8), np.log(250))).astype(np.float32)
    sc = (np.floor(detrand.uniform(83, (n, 1), 0.1, 1) * 16) / 16).astype(np.float32)
    P = np.concatenate([c - s / 2, c + s / 2, sc, detrand.randint(84, (n, 1), 0, 20).astype(np.float32)], 1).astype(np.float32)
    from object_detectors_amd.yolo.utilities import helper
    out = helper.nms_majority(torch.from_numpy(P.copy()).to(dev()), 0.6)
    ref, _ = yo.nms_majority(P, 0.6)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_nms_majority_beyond_one_sort_chunk():
    """helper.nms_majority has no size cap in the reference (helper.py:280-382): 20 000 boxes go through the chunk sorts + merge by rank;
    quantised scores put equal-score runs across the chunk border."""
    n = 20000
    c = detrand.uniform(91, (n, 2), 40, 4000)
    s = np.exp(detrand.uniform(92, (n, 2), np.log(8), np.log(250))).astype(np.float32)
    sc = (np.floor(detrand.uniform(93, (n, 1), 0.1, 1) * 64) / 64).astype(np.float32)
    P = np.concatenate([c - s / 2, c + s / 2, sc, detrand.randint(94, (n, 1), 0, 20).astype(np.float32)], 1).astype(np.float32)
    from object_detectors_amd.yolo.utilities import helper
    out = helper.nms_majority(torch.from_numpy(P.copy()).to(dev()), 0.6)
    ref, _ = yo.nms_majority(P, 0.6)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_postprocess_gpu(golden):
    from object_detectors_amd.yolo.procedures.test_one_epoch import postprocess
    g = golden("g11_postproc")
    g3 = golden("g3_yolo_forw")
    seed, C, img, bs = [int(v) for v in g["meta"]]
    spec = yo.YoloSpec(g3["coco128_anchors"].tolist(), C, img)
    heads = synth_heads(seed, bs, 3, C, (4, 8, 16))
    mod = make_module(spec)
    with torch.no_grad():
        pred = mod([torch.from_numpy(h).to(dev()) for h in heads])
        res = postprocess(pred, float(g["conf"][0]), 0.6)
        # engine-native channels-last heads: fused score/label from the decode kernel
        pred_cl = mod([torch.from_numpy(h).to(dev()).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for h in heads])
        assert mod.last_decode_scores is not None
        res_cl = postprocess(pred_cl, float(g["conf"][0]), 0.6, criterion=mod)
    assert len(res) == 2 and len(res_cl) == 2
    for a, b in zip(res, res_cl):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    for e, fin in enumerate(res):
        fin = fin.cpu().numpy()
        assert fin.shape == g[f"final{e}"].shape
        np.testing.assert_allclose(fin[:, :5], g[f"final{e}"][:, :5], rtol=1e-4, atol=1e-4)
        assert np.array_equal(fin[:, 5], g[f"final{e}"][:, 5])


def test_per_batch_idf_and_config_class_weights(golden, tmp_path):
    """The research knobs of the reference's config (hydra/yolo/head.yaml:18-21): `tfidf_batch` recomputes the logits row from the batch
    (yolo_forw.py:87-91) - same loss as the reference run of that mode (fixture coco128_batchidf) - and `tfidf: [1, 1]` reads class weights
    and the logits row from the cached idf table."""
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    from object_detectors_amd.yolo.utilities.custom import IDFTransformer
    g3 = golden("g3_yolo_forw")
    spec, heads, targets = yolo_case(g3, "coco128_batchidf")
    mod = YOLOForw(anchors=spec.anchors, num_classes=spec.C, img_size=spec.img_size, tfidf_batch=True, tfidf_norm=2).to(dev())
    loss, sub, stats = mod([torch.from_numpy(h).to(dev()) for h in heads], to_targets(targets))
    np.testing.assert_allclose(mod.idf_logits.cpu().numpy(), g3["coco128_batchidf_batch_idf"], rtol=1e-6)
    np.testing.assert_allclose(float(loss), float(g3["coco128_batchidf_loss"]), rtol=1e-4)
    np.testing.assert_allclose(sub.cpu().numpy(), g3["coco128_batchidf_sub_losses"], rtol=1e-4, atol=1e-6)
    # tfidf = [1, 1] from a cached table
    csvp = os.path.join(tmp_path, "idf.csv")
    w = detrand.uniform(5, (80,), 0.5, 2.0)
    with open(csvp, "w") as f:
        f.write("name,smooth,instance_freq\n" + "\n".join(f"c{i},{w[i]:.8f},{100 + i}" for i in range(80)) + "\n")
    idf = IDFTransformer(csv_path=csvp, device="cpu")
    assert idf.num_classes == 80 and set(idf.idf_weights) == {"smooth", "instance_freq"}
    m2 = YOLOForw(anchors=spec.anchors, num_classes=80, img_size=spec.img_size, idf=idf, tfidf=[1, 1], tfidf_norm=0).to(dev())
    np.testing.assert_allclose(m2.class_weights.cpu().numpy(), w, rtol=1e-6)
    np.testing.assert_allclose(m2.idf_logits.cpu().numpy(), w, rtol=1e-6)
    l2, _s, _t = m2([torch.from_numpy(h).to(dev()) for h in heads], to_targets(targets))
    ref = yo.yolo_loss(yo.YoloSpec(spec.anchors, 80, spec.img_size, idf_logits=w, class_weights=w), heads, targets, want_grad=False)
    np.testing.assert_allclose(float(l2), float(ref["loss"]), rtol=2e-4)
    m3 = YOLOForw(anchors=spec.anchors, num_classes=80, img_size=spec.img_size, idf=idf, tfidf=[2, 0]).to(dev())
    freq = 100.0 + np.arange(80)
    we = (1.0 - 0.9999) / (1.0 - np.power(0.9999, freq))
    np.testing.assert_allclose(m3.class_weights.cpu().numpy(), we / we.sum() * 80, rtol=1e-5)
