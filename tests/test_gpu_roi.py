"""GPU parity: RoIAlign / MultiScaleRoIAlign, top-k, RPN proposal filtering and RetinaNet post-processing vs
oracle/tv_oracle.py (torchvision semantics restated; parity unpinned by the reference, see oracle header)."""
import numpy as np
import pytest

from oracle import detrand
from oracle import tv_oracle as tv

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def rand_rois(seed, k, nimg, extent):
    tl = detrand.uniform(seed, (k, 2), 0, extent * 0.7)
    wh = np.exp(detrand.uniform(seed + 1, (k, 2), np.log(4), np.log(extent * 0.6))).astype(np.float32)
    b = detrand.randint(seed + 2, (k, 1), 0, nimg).astype(np.float32)
    return np.concatenate([b, tl, np.minimum(tl + wh, extent)], 1).astype(np.float32)


@pytest.mark.parametrize("aligned,sr", [(False, 2), (True, 2), (False, -1)])
def test_roi_align_single_level(aligned, sr):
    from object_detectors_amd.tvision.roi_align import roi_align
    feat = detrand.uniform(5, (2, 5, 20, 24), -1, 1)
    rois = rand_rois(6, 23, 2, 80.0)
    rois[0, 1:] = [-5, -5, 3, 2]          # partly outside
    rois[1, 1:] = [70, 60, 110, 100]      # beyond the map
    ft = T(feat).requires_grad_(True)
    out = roi_align(ft, T(rois), (7, 7), 0.25, sr, aligned)
    ref = tv.roi_align(feat, rois, (7, 7), 0.25, sr, aligned)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    # backward == adjoint of the forward (linearity property): <g, A f> == <A^T g, f>
    g = detrand.uniform(7, tuple(out.shape), -1, 1)
    out.backward(T(g))
    lhs = float((out.detach().cpu().numpy().astype(np.float64) * g).sum())
    rhs = float((ft.grad.cpu().numpy().astype(np.float64) * feat).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_multiscale_roi_align():
    from object_detectors_amd.tvision.roi_align import MultiScaleRoIAlign
    img = (768, 1024)
    feats = {str(i): detrand.uniform(20 + i, (2, 4, img[0] // s, img[1] // s), -1, 1) for i, s in enumerate((4, 8, 16, 32))}
    feats["pool"] = detrand.uniform(30, (2, 4, 4, 5), -1, 1)       # ignored (not in featmap_names)
    boxes = [rand_rois(40 + i, 30, 1, 760.0)[:, 1:] for i in range(2)]
    m = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    out = m({k: T(v) for k, v in feats.items()}, [T(b) for b in boxes], [img, img]).cpu().numpy()
    allb = np.concatenate(boxes)
    bid = np.concatenate([np.full(len(b), i, np.float32) for i, b in enumerate(boxes)])
    lv = tv.map_levels(allb, 2, 5)
    assert len(set(lv.tolist())) >= 3
    ref = np.zeros_like(out)
    for q, s in enumerate((4, 8, 16, 32)):
        sel = np.nonzero(lv == q)[0]
        if len(sel):
            rois = np.concatenate([bid[sel, None], allb[sel]], 1)
            ref[sel] = tv.roi_align(feats[str(q)], rois, (7, 7), 1.0 / s, 2, False)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=5e-5)   # fp32 kernel vs fp64 oracle


@pytest.mark.parametrize("n,k", [(100, 100), (5000, 1000), (120000, 2000), (40000, 16384)])
def test_topk_rows(n, k):
    from object_detectors_amd import ops
    x = detrand.uniform(50 + n, (3, n), -8, 8)
    x[1, 10:200] = x[1, 5]               # many equal values around the selection boundary
    x[2] = np.round(x[2] * 4) / 4        # heavy ties everywhere
    val, idx, cnt = ops.topk_rows(T(x), k)
    assert cnt.tolist() == [min(k, n)] * 3
    for r in range(3):
        order = np.lexsort((np.arange(n), -x[r].astype(np.float64)))[:k]      # descending value, ties lower index
        assert np.array_equal(idx[r].cpu().numpy(), order)
        assert np.array_equal(val[r].cpu().numpy(), x[r][order])
    # threshold: fewer than k valid
    val, idx, cnt = ops.topk_rows(T(x[:1]), k, min_value=7.5)
    c = int(cnt.item())
    assert c == min(k, int((x[0] > 7.5).sum()))
    assert (val[0, :c].cpu().numpy() > 7.5).all()


def test_rpn_filter_proposals_vs_oracle():
    from object_detectors_amd.tvision.postprocess import rpn_filter_proposals
    levels = [3000, 900, 300]
    A = sum(levels)
    N = 2
    ctr = detrand.uniform(60, (N, A, 2), -20, 420)
    wh = np.exp(detrand.uniform(61, (N, A, 2), np.log(0.5), np.log(200))).astype(np.float32)
    props = np.concatenate([ctr - wh / 2, ctr + wh / 2], 2).astype(np.float32)
    obj = detrand.uniform(62, (N, A), -6, 6)
    shapes = [(400, 380), (360, 400)]
    gb, gs = rpn_filter_proposals(T(props), T(obj), shapes, levels, pre_nms_top_n=600, post_nms_top_n=200, nms_thresh=0.7)
    for i in range(N):
        # numpy restatement of rpn.py:215-280
        idx, lvl, off = [], [], 0
        for li, n in enumerate(levels):
            k = min(600, n)
            o = obj[i, off:off + n]
            top = np.lexsort((np.arange(n), -o.astype(np.float64)))[:k]
            idx.append(top + off)
            lvl.append(np.full(k, li))
            off += n
        idx, lvl = np.concatenate(idx), np.concatenate(lvl)
        sc = (1.0 / (1.0 + np.exp(-obj[i, idx].astype(np.float64)))).astype(np.float32)
        bx = tv.clip_boxes_to_image(props[i, idx], shapes[i])
        keep = tv.remove_small_boxes(bx, 1e-3)
        bx, sc, lvl = bx[keep], sc[keep], lvl[keep]
        keep = tv.batched_nms(bx, sc, lvl, 0.7)[:200]
        got_b, got_s = gb[i].cpu().numpy(), gs[i].cpu().numpy()
        assert got_b.shape == bx[keep].shape
        np.testing.assert_allclose(got_b, bx[keep], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(got_s, sc[keep], rtol=1e-5, atol=1e-6)


def test_retinanet_postprocess_vs_oracle():
    from object_detectors_amd.tvision.postprocess import retinanet_postprocess_detections
    K, N = 11, 2
    grids = [(16, 20), (8, 10), (4, 5)]
    sizes = [(32, 40, 50), (64, 80, 101), (128, 161, 203)]
    anchors = [tv.anchors([s], [(0.5, 1.0, 2.0)], (128, 160), [g]) for s, g in zip(sizes, grids)]
    logits = [detrand.uniform(70 + i, (N, a.shape[0], K), -7, 1) for i, a in enumerate(anchors)]
    regs = [detrand.uniform(80 + i, (N, a.shape[0], 4), -0.5, 0.5) for i, a in enumerate(anchors)]
    tfidf = detrand.uniform(90, (K,), 0.6, 1.6)
    shapes = [(128, 160), (120, 150)]
    det = retinanet_postprocess_detections([T(l) for l in logits], [T(r) for r in regs], [T(a) for a in anchors], shapes,
                                           tfidf_post=T(tfidf), topk_candidates=300, detections_per_img=100)
    for i in range(N):
        ib, isc, il = [], [], []
        for lg, rg, an in zip(logits, regs, anchors):
            s = 1.0 / (1.0 + np.exp(-(lg[i] * tfidf[None, :]).astype(np.float64).reshape(-1)))
            valid = np.nonzero(s > 0.05)[0]
            order = valid[np.lexsort((valid, -s[valid]))][:300]
            a_idx, lab = order // K, order % K
            bx = tv.decode_boxes(rg[i][a_idx], an[a_idx], (1, 1, 1, 1))
            ib.append(tv.clip_boxes_to_image(bx, shapes[i]))
            isc.append(s[order].astype(np.float32))
            il.append(lab)
        b, s, l = np.concatenate(ib), np.concatenate(isc), np.concatenate(il)
        keep = tv.batched_nms(b, s, l, 0.5)[:100]
        assert det[i]["boxes"].shape[0] == len(keep)
        np.testing.assert_allclose(det[i]["boxes"].cpu().numpy(), b[keep], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(det[i]["scores"].cpu().numpy(), s[keep], rtol=1e-5, atol=1e-6)
        assert np.array_equal(det[i]["labels"].cpu().numpy(), l[keep])


def test_roi_align_channels_last_matches_nchw():
    """mi355det_roi_align_nhwc (bf16 NHWC features, lanes over channels) == the NCHW kernel on the same bf16-rounded features,
    forward and backward, multi-level."""
    from object_detectors_amd import ops
    from object_detectors_amd.tvision.roi_align import MultiScaleRoIAlign
    torch.manual_seed(0)
    n, c = 2, 64
    sizes = [(64, 48), (32, 24), (16, 12), (8, 6)]
    img = (256, 192)
    feats_nchw = [torch.randn(n, c, h, w, device="cuda").bfloat16().float() for h, w in sizes]
    feats_cl = [f.permute(0, 2, 3, 1).contiguous().bfloat16() for f in feats_nchw]
    boxes = []
    for i in range(n):
        tl = torch.rand(40, 2, device="cuda") * torch.tensor([150.0, 100.0], device="cuda")
        wh = torch.rand(40, 2, device="cuda") * 120 + 4
        boxes.append(torch.cat([tl, tl + wh], 1))
    pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    fa = [f.clone().requires_grad_(True) for f in feats_nchw]
    fb = [f.clone().requires_grad_(True) for f in feats_cl]
    from collections import OrderedDict
    ya = pool(OrderedDict((str(i), f) for i, f in enumerate(fa)), boxes, [img] * n)
    yb = pool.forward_nhwc(fb, boxes, [img] * n)
    assert ya.shape == yb.shape == (80, c, 7, 7)
    torch.testing.assert_close(yb, ya, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    for a, b in zip(fa, fb):
        want = a.grad.permute(0, 2, 3, 1)
        assert b.grad.dtype == torch.bfloat16
        assert float((b.grad.float() - want).abs().max()) <= 1e-2 * float(want.abs().max()) + 1e-6


def test_topk_long_rows_multi_workgroup():
    """mi355det_topk_ws (several workgroups per row) against a numpy selection: random rows, a threshold that cuts the candidates,
    fewer valid elements than k, and rows of heavily repeated values (ties at the threshold, incl. the overflow fallback)."""
    from object_detectors_amd import ops
    rng = np.random.default_rng(3)

    def ref_topk(row, k, min_value):
        order = np.lexsort((np.arange(row.size), -row.astype(np.float64)))        # descending value, ties -> lower index
        order = order[row[order] > min_value][:k]
        return order

    cases = [
        (rng.standard_normal((2, 300000)).astype(np.float32), 2000, -np.inf),
        (rng.standard_normal((3, 70001)).astype(np.float32), 1000, 1.5),            # threshold leaves more than k
        (rng.standard_normal((1, 200000)).astype(np.float32), 1000, 3.9),           # fewer than k valid
        (np.round(rng.standard_normal((2, 150000)) * 4).astype(np.float32) / 4, 3000, -np.inf),    # many exact ties at the cut
        (np.zeros((1, 100000), np.float32), 500, -1.0),                              # everything ties: candidate list overflows -> exact redo
        (np.concatenate([np.full((1, 90000), 2.0, np.float32), rng.standard_normal((1, 10000)).astype(np.float32)], 1), 700, -np.inf),
    ]
    for x, k, mv in cases:
        val, idx, cnt = ops.topk_rows(torch.from_numpy(x).to("cuda:0"), k, min_value=float(mv) if np.isfinite(mv) else float("-inf"))
        val, idx, cnt = val.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
        for r in range(x.shape[0]):
            want = ref_topk(x[r], k, mv)
            assert cnt[r] == want.size, (x.shape, k, r, cnt[r], want.size)
            assert np.array_equal(idx[r, :cnt[r]], want), (x.shape, k, r)
            assert np.array_equal(val[r, :cnt[r]], x[r][want])


@pytest.mark.parametrize("sampling,aligned", [(2, False), (2, True), (0, False), (3, True)])
def test_roi_align_separable_edge_cases(sampling, aligned):
    """The separable 7x7 form of mi355det_roi_align_nhwc (footprint weights per axis) against the per-sample NCHW kernel (pinned by the
    oracle above) on boxes that leave the map, degenerate boxes, whole-image and very elongated boxes; forward and backward; 300 channels
    (more than one pass of the channel loop)."""
    from object_detectors_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(4 + sampling)
    n, c = 2, 300
    sizes, scales = [(50, 68), (25, 34), (13, 17), (7, 9)], [0.25, 0.125, 0.0625, 0.03125]
    feats = [torch.randn((n, c, h, w), device=dev, generator=g).bfloat16().float() for h, w in sizes]
    feats_cl = [f.permute(0, 2, 3, 1).contiguous().bfloat16() for f in feats]
    special = torch.tensor([[-40.0, -30.0, 20.0, 25.0], [250.0, 180.0, 330.0, 260.0], [100.0, 100.0, 100.0, 100.0], [0.0, 0.0, 272.0, 200.0],
                            [5.0, 90.0, 270.0, 96.0], [130.0, 2.0, 134.0, 198.0], [300.0, 300.0, 400.0, 400.0], [-500.0, -500.0, -400.0, -450.0],
                            [10.0, 10.0, 11.5, 12.0], [60.0, 40.0, 200.0, 190.0]], device=dev)
    tl = torch.rand((50, 2), device=dev, generator=g) * torch.tensor([220.0, 160.0], device=dev)
    wh = torch.rand((50, 2), device=dev, generator=g) ** 2 * 200 + 1
    boxes = torch.cat([special, torch.cat([tl, tl + wh], 1)])
    rois = torch.cat([(torch.arange(boxes.shape[0], device=dev) % n).float()[:, None], boxes], 1)
    ya = ops.roi_align_multi(feats, rois, 7, scales, sampling, aligned, 2, 5)
    yb = ops.roi_align_nhwc(feats_cl, rois, 7, scales, sampling, aligned, 2, 5)
    torch.testing.assert_close(yb, ya, rtol=1e-5, atol=2e-5)
    go = torch.randn(ya.shape, device=dev, generator=g)
    da = ops.roi_align_multi(feats, rois, 7, scales, sampling, aligned, 2, 5, grad_out=go)
    db = ops.roi_align_nhwc(feats_cl, rois, 7, scales, sampling, aligned, 2, 5, grad_out=go)
    for a, b_ in zip(da, db):
        torch.testing.assert_close(b_, a.permute(0, 2, 3, 1), rtol=1e-4, atol=1e-4)
