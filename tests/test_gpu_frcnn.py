"""Faster R-CNN training-side mirrors (tvision/rpn.py, tvision/roi_heads.py) on the GPU against the reference fixture g13 and the
oracle: bit-exact labels / matched indices, losses within 1e-4."""
import numpy as np
import pytest

from oracle import tv_oracle as tv

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def dev():
    return torch.device("cuda:0")


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
    with pytest.raises(ValueError):
        fastrcnn_loss(T(g["frcnn_logits"]), T(g["frcnn_breg"]), [T(g["frcnn_labels"])], [T(g["frcnn_tgt"])], loss_type="hinge")


def test_fastrcnn_loss_kernel_all_variants_vs_reference(golden):
    """mi355det_fastrcnn_loss (fused forward + gradients) against the reference's own fastrcnn_loss outputs and autograd gradients for every
    loss_type, called as RoIHeads.forward does (roi_heads.py:826-827): tf-idf row on the logits, class weights for 'ce'; both sides of the
    gombit "/4 above 5" branch."""
    from object_detectors_amd.tvision.roi_heads import fastrcnn_loss
    from tests.test_oracle_frcnn import LOSS_CASES, loss_case_inputs
    g = golden("g13_frcnn")
    tfidf, cw = T(g["frcnn_tfidf"]), T(g["frcnn_cw"])
    for lt, scale in LOSS_CASES:
        logits_np, tag = loss_case_inputs(g, lt, scale)
        lg = T(logits_np).requires_grad_(True)
        br = T(g["frcnn_breg"]).requires_grad_(True)
        c, b = fastrcnn_loss(lg, br, [T(g["frcnn_labels"])], [T(g["frcnn_tgt"])], weights=cw if lt == "ce" else None, loss_type=lt, class_scale=tfidf)
        (c + b).backward()
        np.testing.assert_allclose([float(c), float(b)], g[f"frcnn_w_losses_{tag}"], rtol=1e-4, err_msg=tag)
        ref = g[f"frcnn_w_glogits_{tag}"]
        np.testing.assert_allclose(lg.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-5 * float(np.abs(ref).max()), err_msg=tag)
        if lt == "ce":
            np.testing.assert_allclose(br.grad.cpu().numpy(), g["frcnn_w_gbreg"], rtol=1e-5, atol=1e-8)
        # pre-multiplied logits, as the reference's call site does it, give the same losses
        c2, b2 = fastrcnn_loss(tfidf * T(logits_np), T(g["frcnn_breg"]), [T(g["frcnn_labels"])], [T(g["frcnn_tgt"])],
                               weights=cw if lt == "ce" else None, loss_type=lt)
        np.testing.assert_allclose([float(c2), float(b2)], [float(c), float(b)], rtol=1e-5)


def test_fastrcnn_loss_lvis_width_vs_oracle():
    """K = 1204 (LVIS, BASELINE config 5 head width), 1024 rows: losses and gradients against the float64 oracle."""
    from object_detectors_amd import ops
    from oracle import detrand
    n, k = 1024, 1204
    x = detrand.uniform(51, (n, k), -4, 4)
    br = detrand.uniform(52, (n, 4 * k), -1, 1)
    lab = detrand.randint(53, (n,), 0, k)
    lab[::2] = 0
    tgt = detrand.uniform(54, (n, 4), -1, 1)
    sc = detrand.uniform(55, (k,), 0.5, 1.5)
    for lt in ("ce", "focal_loss", "gombit_fl"):
        losses, gl, gb = ops.fastrcnn_loss(T(x), T(br), T(lab), T(tgt), class_scale=T(sc), loss_type=lt)
        oc, ob, ogl, ogb = tv.fastrcnn_loss(sc[None, :] * x, br, lab, tgt, lt, want_grad=True)
        np.testing.assert_allclose(losses.cpu().numpy(), [oc, ob], rtol=1e-4)
        np.testing.assert_allclose(gl.cpu().numpy(), ogl * sc[None, :], rtol=2e-3, atol=1e-5 * float(np.abs(ogl).max()))
        np.testing.assert_allclose(gb.cpu().numpy(), ogb, rtol=1e-5, atol=1e-9)


def test_roi_postprocess_detections_vs_reference(golden):
    """tvision/postprocess.py:roi_heads_postprocess_detections against RoIHeads.postprocess_detections of the reference (fixture g13: softmax,
    sigmoid and gombit scores, tf-idf row, background removal, score / small-box filters, per-class NMS, top-k)."""
    from object_detectors_amd.tvision.postprocess import roi_heads_postprocess_detections
    g = golden("g13_frcnn")
    props = [T(g["pp_props0"]), T(g["pp_props1"])]
    for lt in ("ce", "bce", "gombit"):
        b, s, l = roi_heads_postprocess_detections(T(g["pp_logits"]), T(g["pp_breg"]), props, [(512, 640), (480, 512)], T(g["pp_tfidf_post"]), 0.05, 0.5, 20,
                                                   (10.0, 10.0, 5.0, 5.0), lt)
        for i in range(2):
            assert np.array_equal(l[i].cpu().numpy(), g[f"pp_{lt}_labels{i}"]), (lt, i)
            np.testing.assert_allclose(s[i].cpu().numpy(), g[f"pp_{lt}_scores{i}"], rtol=1e-5)
            np.testing.assert_allclose(b[i].cpu().numpy(), g[f"pp_{lt}_boxes{i}"], rtol=1e-5, atol=1e-3)


def _per_class_nms_topn(boxes, scores, labels, thr, top_n):
    """independent statement of batched_nms(...)[:top_n]: NMS inside every class, survivors merged by (score desc, index asc)."""
    keep = []
    for c in np.unique(labels):
        idx = np.nonzero(labels == c)[0]
        keep.append(idx[tv.nms(boxes[idx], scores[idx], thr)])
    keep = np.concatenate(keep)
    order = np.lexsort((keep, -scores[keep].astype(np.float64)))
    return keep[order][:top_n]


def test_batched_nms_topn_without_candidate_cap():
    """More candidates than one NMS launch holds (the LVIS shape: 1203 classes): the chunked form is EXACTLY batched_nms(...)[:top_n]."""
    from object_detectors_amd.tvision.postprocess import batched_nms_topn
    from oracle import detrand
    n, ncls = 40000, 1203
    c = detrand.uniform(61, (n, 2), 0, 760)
    wh = np.exp(detrand.uniform(62, (n, 2), np.log(8), np.log(300))).astype(np.float32)
    boxes = np.concatenate([c, c + wh], 1).astype(np.float32)
    scores = detrand.uniform(63, (n,), 0.05, 1.0)
    labels = detrand.randint(64, (n,), 1, ncls + 1)
    for top_n in (100, 300):
        kb, ks, kl = batched_nms_topn(T(boxes), T(scores), T(labels), 0.5, top_n)
        want = _per_class_nms_topn(boxes, scores, labels, 0.5, top_n)
        assert np.array_equal(ks.cpu().numpy(), scores[want]) and np.array_equal(kl.cpu().numpy(), labels[want])
        assert np.array_equal(kb.cpu().numpy(), boxes[want])
    # several chunks: few classes and heavy overlap leave fewer than top_n survivors per chunk (capacity lowered to force it)
    n2 = 3000
    boxes2 = boxes[:n2] * np.float32(0.15)
    labels2 = detrand.randint(65, (n2,), 1, 4)
    for top_n in (5, 60, 400):
        kb, ks, kl = batched_nms_topn(T(boxes2), T(scores[:n2]), T(labels2), 0.5, top_n, capacity=512)
        want = tv.batched_nms(boxes2, scores[:n2], labels2, 0.5)[:top_n]
        assert np.array_equal(ks.cpu().numpy(), scores[:n2][want]) and np.array_equal(kl.cpu().numpy(), labels2[want])


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


@pytest.mark.parametrize("rows,fin,fout,relu", [(300, 12544, 1024, True), (2048, 1024, 1024, True), (257, 1024, 91, False), (512, 1024, 364, False)])
def test_mfma_linear_matches_fp32_linear(rows, fin, fout, relu):
    """Box-head layers (frcnn.py:243-289) on the library's MFMA kernels against F.linear in fp32 on the same bf16-rounded operands:
    output, input gradient, weight and bias gradients; parameter names / shapes are nn.Linear's."""
    import torch.nn.functional as F
    from object_detectors_amd.tvision.linear import MfmaLinear
    torch.manual_seed(rows + fout)
    m = MfmaLinear(fin, fout, relu=relu).to(dev())
    assert tuple(m.weight.shape) == (fout, fin) and tuple(m.bias.shape) == (fout,) and set(dict(m.named_parameters())) == {"weight", "bias"}
    x = (torch.randn(rows, fin, device=dev()) * 0.5).bfloat16().float().requires_grad_(True)
    g = torch.randn(rows, fout, device=dev()).bfloat16().float()
    y = m(x)
    y.backward(g)
    wq = m.weight.detach().bfloat16().float().requires_grad_(True)
    bq = m.bias.detach().clone().requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.linear(xr, wq, bq)
    if relu:
        yr = torch.relu(yr)
    yr.backward(g)
    rel = lambda a, b: float((a.float() - b.float()).abs().max()) / (float(b.float().abs().max()) + 1e-30)
    assert rel(y, yr) < 1.5e-2
    assert rel(x.grad, xr.grad) < 2e-2
    assert rel(m.weight.grad, wq.grad) < 2e-2 and rel(m.bias.grad, bq.grad) < 2e-2
    # a parameter update invalidates the cached packs
    with torch.no_grad():
        m.weight.mul_(2.0)
    y2 = m(x.detach())
    assert rel(y2, torch.relu(F.linear(x.detach(), wq.detach() * 2, bq.detach())) if relu else F.linear(x.detach(), wq.detach() * 2, bq.detach())) < 1.5e-2
