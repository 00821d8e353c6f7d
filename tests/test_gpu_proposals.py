"""mi355det_rpn_proposals (one host call for RegionProposalNetwork.filter_proposals, tvision/rpn.py:215-280 + the decode of :336-351)
against the composed route (ops.box_decode of every anchor + postprocess.rpn_filter_proposals, itself pinned by the oracle in
tests/test_gpu_frcnn.py): bit for bit."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(seed, n, levels, sizes):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    a = sum(levels)
    obj = torch.randn((n, a), device=dev, generator=g) * 3
    ctr = torch.rand((a, 2), device=dev, generator=g) * 900 - 50            # some anchors partly / fully outside the image
    wh = torch.rand((a, 2), device=dev, generator=g) * 300 + 2
    anchors = torch.cat([ctr - wh / 2, ctr + wh / 2], -1)
    deltas = torch.randn((n, a, 4), device=dev, generator=g) * 0.5
    deltas[:, ::97, 2:] = 9.0                                               # beyond bbox_xform_clip
    deltas[:, ::53, 2:] = -20.0                                             # collapses to a box below min_size
    return obj, deltas, anchors, sizes[:n]


@pytest.mark.parametrize("n,levels,pre,post,score_thresh", [
    (4, [200 * 200 * 3, 100 * 100 * 3, 50 * 50 * 3, 25 * 25 * 3, 13 * 13 * 3], 2000, 2000, 0.0),       # training, 800 px
    (2, [200 * 200 * 3, 100 * 100 * 3, 50 * 50 * 3, 25 * 25 * 3, 13 * 13 * 3], 1000, 1000, 0.0),       # testing
    (3, [300, 75, 21], 1000, 50, 0.0),                                                                 # k = whole level, cut at post
    (2, [5000, 1200], 600, 300, 0.6),                                                                  # score threshold
    (1, [4096], 4096, 1000, 0.0),
])
def test_rpn_proposals_matches_composed_route(n, levels, pre, post, score_thresh):
    from object_detectors_amd import ops
    from object_detectors_amd.tvision.postprocess import rpn_filter_proposals, rpn_proposals_fused
    sizes = [(800, 800), (640, 768), (512, 800), (800, 600)]
    obj, deltas, anchors, shapes = _inputs(5 + n + len(levels), n, levels, sizes)
    clip = math.log(1000.0 / 16)
    props = ops.box_decode(deltas.reshape(-1, 4), anchors.repeat(n, 1), (1.0, 1.0, 1.0, 1.0), clip).reshape(n, -1, 4)
    rb, rs = rpn_filter_proposals(props, obj, shapes, levels, pre, post, 0.7, score_thresh)
    fb, fs = rpn_proposals_fused(deltas, obj, anchors, shapes, levels, pre, post, 0.7, score_thresh, xform_clip=clip)
    assert len(fb) == len(rb) == n
    for i in range(n):
        assert fb[i].shape == rb[i].shape and fb[i].shape[0] > 0, (i, fb[i].shape, rb[i].shape)
        assert torch.equal(fb[i], rb[i]), (i, (fb[i] - rb[i]).abs().max())
        assert torch.equal(fs[i], rs[i]), (i, (fs[i] - rs[i]).abs().max())


def test_rpn_proposals_rows_with_few_finite_logits_and_bad_arguments():
    from object_detectors_amd import ops
    dev = torch.device("cuda:0")
    obj, deltas, anchors, _ = _inputs(3, 2, [900, 100], [(400, 400)] * 2)
    obj[0, :880] = float("-inf")                               # fewer than k finite logits in level 0 of image 0
    lim = torch.tensor([[400.0, 400.0, 400.0, 400.0]] * 2, device=dev)
    boxes, scores, counts = ops.rpn_proposals(obj, deltas, anchors, lim, [900, 100], 500, 200, 0.7)
    c = counts.tolist()
    assert 0 < c[0] <= 120 and 0 < c[1] <= 200
    assert torch.isfinite(boxes).all() and (boxes[0, c[0]:] == 0).all() and (scores[0, c[0]:] == 0).all()
    assert (boxes >= 0).all() and (boxes <= 400).all()
    with pytest.raises(ValueError):
        ops.rpn_proposals(obj, deltas, anchors, lim, [900, 99], 500, 200, 0.7)          # levels do not add up to A
    with pytest.raises(ValueError):
        ops.rpn_proposals(obj, deltas, anchors, lim, [100] * 10, 500, 200, 0.7)         # more than 8 levels


@pytest.mark.parametrize("rows,segs,k", [
    (4, [120000, 30000, 7500, 1875, 507], 2000),       # the RPN pyramid at 800 px (first segment: the many-workgroup route)
    (3, [30001, 4099, 63], 1000),                      # unaligned starts, k = whole segment
    (2, [65535, 17], 16384),                           # the longest short segment, the largest k
    (5, [9000], 300),
])
def test_topk_segments_matches_topk_rows(rows, segs, k):
    from object_detectors_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(rows * 7 + len(segs))
    x = torch.randn((rows, sum(segs)), device=dev, generator=g)
    x[0, : min(5000, segs[0])] = 0.25                  # a run of identical keys around / above the threshold
    if rows > 1:
        x[1] = torch.round(x[1] * 4) / 4               # heavy ties everywhere
    outs = ops.topk_segments(x, segs, k)
    off = 0
    for (val, idx, cnt), n in zip(outs, segs):
        rv, ri, rc = ops.topk_rows(x[:, off:off + n], k)
        assert torch.equal(cnt, rc) and torch.equal(idx, ri) and torch.equal(val, rv), (n, (idx != ri).sum().item())
        tv, ti = torch.sort(x[:, off:off + n], dim=1, descending=True, stable=True)
        kk = min(k, n)
        assert torch.equal(val, tv[:, :kk]) and torch.equal(idx, ti[:, :kk])
        off += n


def test_topk_segments_degenerate_rows_and_threshold():
    from object_detectors_amd import ops
    dev = torch.device("cuda:0")
    x = torch.zeros((3, 40000 + 20000), device=dev)
    x[0] = 1.5                                          # every key identical: more ties than the candidate list holds -> ordered fallback
    x[1, ::3] = float("-inf")
    x[1, 1::3] = torch.arange(20000, device=dev, dtype=torch.float32)
    x[2] = torch.arange(60000, device=dev, dtype=torch.float32) % 7
    outs = ops.topk_segments(x, [40000, 20000], 3000, min_value=0.5)
    off = 0
    for (val, idx, cnt), n in zip(outs, [40000, 20000]):
        rv, ri, rc = ops.topk_rows(x[:, off:off + n], 3000, min_value=0.5)
        assert torch.equal(cnt, rc)
        for r in range(3):
            c = int(cnt[r])
            assert torch.equal(idx[r, :c], ri[r, :c]) and torch.equal(val[r, :c], rv[r, :c])
        off += n


@pytest.mark.parametrize("n,post,gts", [(4, 2000, [3, 1, 17, 6]), (2, 300, [40, 2]), (1, 64, [5])])
def test_select_training_samples_fused_matches_composed_route(n, post, gts):
    """mi355det_roi_match + mi355det_roi_sample against the per-image torch-composed RoIHeadTargets.select_training_samples
    (roi_heads.py:664-713) under the same generator state: same samples in the same order, bit-equal targets."""
    from object_detectors_amd.tvision.roi_heads import RoIHeadTargets
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11 + n)
    targets, boxes, counts = [], [], []
    pad = torch.zeros((n, post, 4), device=dev)
    for i in range(n):
        c = torch.rand((gts[i], 2), device=dev, generator=g) * 600 + 100
        wh = torch.rand((gts[i], 2), device=dev, generator=g) * 200 + 20
        gt = torch.cat([c - wh / 2, c + wh / 2], 1)
        targets.append({"boxes": gt, "labels": torch.randint(1, 91, (gts[i],), device=dev, generator=g)})
        cnt = post - 7 * i                                                   # ragged proposal counts
        # proposals: jittered copies of the ground truth (positives), boxes elsewhere (negatives), a few between the thresholds
        src = gt[torch.randint(0, gts[i], (cnt,), device=dev, generator=g)]
        jit = torch.randn((cnt, 4), device=dev, generator=g) * torch.rand((cnt, 1), device=dev, generator=g) * 60
        p = src + jit
        p = torch.cat([torch.minimum(p[:, :2], p[:, 2:] - 1), torch.maximum(p[:, 2:], p[:, :2] + 1)], 1).clamp(0, 800)
        boxes.append(p)
        pad[i, :cnt] = p
        counts.append(cnt)
    tg = RoIHeadTargets()
    assert tg.fused_ok(n, post, targets)
    torch.manual_seed(1234)
    r_props, r_mi, r_lab, r_reg = tg.select_training_samples([b.clone() for b in boxes], targets)
    meta = torch.zeros(3 * n, device=dev, dtype=torch.int32)
    meta[:n] = torch.tensor(counts, dtype=torch.int32)
    torch.manual_seed(1234)
    rois, mi, lab, reg, per_image = tg.select_training_samples_fused(pad, meta, targets)
    assert per_image == [int(p.shape[0]) for p in r_props] and sum(per_image) == rois.shape[0]
    assert (torch.cat(r_lab) >= 1).any() and (torch.cat(r_lab) == 0).any()
    ids = torch.cat([torch.full((k,), float(i), device=dev) for i, k in enumerate(per_image)])
    assert torch.equal(rois[:, 0], ids)
    assert torch.equal(rois[:, 1:], torch.cat(r_props))
    assert torch.equal(mi, torch.cat(r_mi)) and torch.equal(lab, torch.cat(r_lab))
    assert torch.equal(reg, torch.cat(r_reg)), (reg - torch.cat(r_reg)).abs().max()


@pytest.mark.parametrize("n,k_cls,prior,tfidf", [(3, 91, 0.05, False), (2, 20, 0.01, True), (2, 1204, 0.05, False)])
def test_retina_detections_matches_composed_route(n, k_cls, prior, tfidf, monkeypatch):
    """mi355det_retina_detections (RetinaNet.postprocess_detections, retinanet.py:414-472, one host call) against the torch-composed chain of
    postprocess.retinanet_postprocess_detections (pinned by the oracle in tests/test_gpu_roi.py): identical detections."""
    from object_detectors_amd.tvision import postprocess as pp
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(21 + k_cls)
    sides = [(40, 40), (20, 20), (10, 10), (5, 5), (3, 3)]
    logits, regs, anchors = [], [], []
    bias = math.log(prior / (1 - prior))
    for li, (h, w) in enumerate(sides):
        hwa = h * w * 9
        logits.append(torch.randn((n, hwa, k_cls), device=dev, generator=g) + bias)
        regs.append(torch.randn((n, hwa, 4), device=dev, generator=g) * 0.3)
        c = torch.rand((hwa, 2), device=dev, generator=g) * 320
        wh = torch.rand((hwa, 2), device=dev, generator=g) * (20 * 2 ** li) + 4
        anchors.append(torch.cat([c - wh / 2, c + wh / 2], 1))
    logits[4][1] = -20.0                                    # a level / image without a single score above the threshold
    shapes = [(320, 320), (300, 256), (256, 320)][:n]
    tf = (torch.rand((1, k_cls), device=dev, generator=g) + 0.5) if tfidf else None
    monkeypatch.setattr(pp, "_RETINA_FUSED", False)
    ref = pp.retinanet_postprocess_detections(logits, regs, anchors, shapes, tf, 0.05, 1000, 0.5, 300)
    monkeypatch.setattr(pp, "_RETINA_FUSED", True)
    got = pp.retinanet_postprocess_detections(logits, regs, anchors, shapes, tf, 0.05, 1000, 0.5, 300)
    assert sum(int(d["boxes"].shape[0]) for d in ref) > 0
    for r, d in zip(ref, got):
        assert d["boxes"].shape == r["boxes"].shape
        assert torch.equal(d["boxes"], r["boxes"]) and torch.equal(d["scores"], r["scores"]) and torch.equal(d["labels"], r["labels"])


def test_rpn_loss_kernel_matches_autograd():
    """mi355det_rpn_loss against RPNTargets.losses_prepared under autograd (rpn.py:282-318): losses and both gradients."""
    from object_detectors_amd.tvision.rpn import RPNTargets
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9)
    t = 4 * 159882
    obj = torch.randn((t, 1), device=dev, generator=g) * 3
    dl = torch.randn((t, 4), device=dev, generator=g) * 0.3
    perm = torch.randperm(t, device=dev, generator=g)
    pos, neg = perm[:317].sort().values, perm[317:1024].sort().values
    labels = torch.zeros(t, device=dev)
    labels[pos] = 1.0
    reg = torch.zeros((t, 4), device=dev)
    reg[pos] = dl[pos] + torch.randn((317, 4), device=dev, generator=g) * 0.2      # both smooth-L1 branches (beta = 1/9)
    reg[pos[:5]] = dl[pos[:5]]                                                      # zero difference
    prep = dict(pos=pos, sampled=torch.cat([pos, neg]), labels=labels, reg=reg)
    tg = RPNTargets()
    o, d = obj.clone().requires_grad_(True), dl.clone().requires_grad_(True)
    ref = tg.losses_prepared(o, d, prep)
    (ref["loss_objectness"] + ref["loss_rpn_box_reg"]).backward()
    got, g_obj, g_dl = tg.losses_prepared_fused(obj, dl, prep)
    torch.testing.assert_close(got["loss_objectness"], ref["loss_objectness"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got["loss_rpn_box_reg"], ref["loss_rpn_box_reg"], rtol=1e-5, atol=1e-6)
    assert g_obj.shape == o.grad.shape and g_dl.shape == d.grad.shape
    torch.testing.assert_close(g_obj, o.grad, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(g_dl, d.grad, rtol=1e-5, atol=1e-9)
    assert int((g_obj != 0).sum()) <= 1024 and int((g_dl != 0).sum()) <= 4 * 317


@pytest.mark.parametrize("loss_type,c", [("ce", 91), ("bce", 21), ("gombit", 91)])
def test_roi_detections_batch_matches_per_image_route(loss_type, c):
    """mi355det_roi_detections (RoIHeads.postprocess_detections, roi_heads.py:715-781, padded proposals, one host call) against the per-image
    composed route (pinned by the oracle in tests/test_gpu_roi.py): identical detections; saturation reported as None."""
    from object_detectors_amd.tvision.postprocess import roi_heads_postprocess_detections, roi_heads_postprocess_detections_batch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(31 + c)
    n, p = 3, 300
    counts = [300, 257, 12]
    shapes = [(400, 500), (384, 512), (400, 400)]
    ctr = torch.rand((n, p, 2), device=dev, generator=g) * 380 + 10
    wh = torch.rand((n, p, 2), device=dev, generator=g) * 150 + 2
    pad = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).clamp(0, 512)
    for i, k in enumerate(counts):
        pad[i, k:] = 0
    scale = 3.0 if loss_type == "ce" else 1.5
    logits = torch.randn((n * p, c), device=dev, generator=g) * scale - (0.0 if loss_type == "ce" else 2.5)
    reg = torch.randn((n * p, c * 4), device=dev, generator=g) * 0.5
    reg[::7, 2::4] = -60.0                                        # collapsed boxes: dropped by remove_small_boxes
    tf = torch.rand((1, c), device=dev, generator=g) + 0.5
    cnt_dev = torch.tensor(counts, device=dev, dtype=torch.int32)
    rows = torch.cat([torch.arange(k, device=dev) + i * p for i, k in enumerate(counts)])
    ref = roi_heads_postprocess_detections(logits[rows], reg[rows], [pad[i, :k] for i, k in enumerate(counts)], shapes, tf, 0.05, 0.5, 100,
                                           (10.0, 10.0, 5.0, 5.0), loss_type)
    got = roi_heads_postprocess_detections_batch(logits, reg, pad, cnt_dev, shapes, tf, 0.05, 0.5, 100, (10.0, 10.0, 5.0, 5.0), loss_type)
    assert got is not None and sum(int(b.shape[0]) for b in ref[0]) > 20
    for q in range(3):
        for a, b in zip(ref[q], got[q]):
            assert a.shape == b.shape and torch.equal(a, b), (q, a.shape, b.shape)
    assert roi_heads_postprocess_detections_batch(logits, reg, pad, cnt_dev, shapes, tf, 0.05, 0.5, 100, (10.0, 10.0, 5.0, 5.0), loss_type,
                                                  max_candidates=16) is None
