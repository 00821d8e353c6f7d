"""GPU parity for the torchvision_models-side kernels vs oracle/tv_oracle.py and reference fixtures."""
import numpy as np
import pytest

from oracle import detrand
from oracle import tv_oracle as tv
from tests.test_oracle_tv import ANCHOR_TAGS, build_anchors

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.mark.parametrize("tag", ANCHOR_TAGS)
def test_anchor_generator_gpu(golden, tag):
    from object_detectors_amd.tvision.anchor_utils import AnchorGenerator
    g = golden("g5_7_tvision")
    sizes = tuple(tuple(r) for r in g[f"anc_{tag}_sizes"].tolist())
    ars = tuple(tuple(r) for r in g[f"anc_{tag}_ars"].tolist())
    img = tuple(int(v) for v in g[f"anc_{tag}_img"])
    grids = [tuple(int(v) for v in r) for r in g[f"anc_{tag}_grids"]]

    class IL:
        tensors = torch.zeros(2, 3, *img)
        image_sizes = [img, img]
    ag = AnchorGenerator(sizes, ars)
    out = ag(IL, [torch.zeros(2, 1, h, w, device=dev()) for h, w in grids])
    a = out[0].cpu().numpy()
    assert np.array_equal(a, build_anchors(g, tag))           # oracle, bit-exact
    assert np.array_equal(a[:64], g[f"anc_{tag}_head"]) and np.array_equal(a[::1009], g[f"anc_{tag}_sample"])


def test_box_iou_gpu():
    a = np.concatenate([detrand.uniform(1, (37, 2), 0, 500), detrand.uniform(2, (37, 2), 500, 900)], 1)
    b = np.concatenate([detrand.uniform(3, (1000, 2), 0, 600), detrand.uniform(4, (1000, 2), 300, 900)], 1)
    b[5] = b[4]
    from object_detectors_amd.tvision import boxes
    got = boxes.box_iou(T(a), T(b)).cpu().numpy()
    assert np.array_equal(got, tv.box_iou(a, b), equal_nan=True)


@pytest.mark.parametrize("tag,anc", [("retina", "retina800"), ("rpn", "frcnn800"), ("roi", None),
                                     ("retina_m1", "retina800"), ("retina_m20", "retina800")])
def test_match_anchors_gpu(golden, tag, anc):
    from object_detectors_amd.tvision._utils import Matcher
    g = golden("g5_7_tvision")
    anchors = g["match_roi_anchors"] if anc is None else build_anchors(g, anc)
    hi, lo, lowq = g[f"match_{tag}_cfg"]
    m = Matcher(float(hi), float(lo), bool(lowq)).match_boxes(T(g[f"match_{tag}_gt"]), T(anchors)).cpu().numpy()
    nz = np.nonzero(m != -1)[0]
    assert np.array_equal(nz, g[f"match_{tag}_nz_idx"])          # reference fixture, bit-exact
    assert np.array_equal(m[nz], g[f"match_{tag}_nz_val"])


def test_matcher_errors_gpu():
    from object_detectors_amd.tvision._utils import Matcher
    with pytest.raises(ValueError):
        Matcher(0.5, 0.4, True).match_boxes(torch.zeros(0, 4, device=dev()), torch.zeros(5, 4, device=dev()))
    with pytest.raises(ValueError):
        Matcher(0.5, 0.4, True)(torch.zeros(0, 5, device=dev()))


@pytest.mark.parametrize("tag", ["w1", "w10"])
def test_box_coder_gpu(golden, tag):
    from object_detectors_amd.tvision._utils import BoxCoder
    g = golden("g5_7_tvision")
    bc = BoxCoder(tuple(g[f"coder_{tag}_w"].tolist()))
    enc = bc.encode_single(T(g[f"coder_{tag}_ref"]), T(g[f"coder_{tag}_prop"])).cpu().numpy()
    np.testing.assert_allclose(enc, g[f"coder_{tag}_enc"], rtol=1e-5, atol=1e-6)
    dec = bc.decode_single(T(g[f"coder_{tag}_codes"]), T(g[f"coder_{tag}_prop"])).cpu().numpy()
    np.testing.assert_allclose(dec, g[f"coder_{tag}_dec"], rtol=1e-5, atol=1e-3)
    dec3 = bc.decode_single(T(g[f"coder_{tag}_codes3"]), T(g[f"coder_{tag}_prop"])).cpu().numpy()
    np.testing.assert_allclose(dec3, g[f"coder_{tag}_dec3"], rtol=1e-5, atol=1e-3)


def nms_inputs(seed, n, extent=800.0):
    c = detrand.uniform(seed, (n, 2), 0, extent)
    s = np.exp(detrand.uniform(seed + 1, (n, 2), np.log(8), np.log(400))).astype(np.float32)
    boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    scores = detrand.uniform(seed + 2, (n,), 0, 1)
    return boxes, scores


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 5000, 10000])
def test_nms_gpu(n):
    from object_detectors_amd.tvision import boxes as B
    boxes, scores = nms_inputs(300 + n, n)
    if n >= 64:
        scores[10] = scores[20]     # equal scores: lower index first
    keep = B.nms(T(boxes), T(scores), 0.5).cpu().numpy()
    assert np.array_equal(keep, tv.nms(boxes, scores, 0.5))


@pytest.mark.parametrize("n", [4097, 6000, 16384])
def test_nms_many_equal_scores_radix_path(n):
    """n >= 4096 takes the radix sort of nms_sort_kernel: scores quantised to 32 levels, so almost every box has equal-score neighbours and
    the keep list depends on the tie order (descending score, lower index first)."""
    from object_detectors_amd.tvision import boxes as B
    boxes, scores = nms_inputs(4000 + n, n, extent=3000.0)
    scores = (np.floor(scores * 32) / 32).astype(np.float32)
    keep = B.nms(T(boxes), T(scores), 0.5).cpu().numpy()
    assert np.array_equal(keep, tv.nms(boxes, scores, 0.5))


@pytest.mark.parametrize("n", [16385, 40000])
def test_nms_beyond_one_sort_chunk(n):
    """More than 16 384 boxes (the reference's nms has no cap): chunk sorts in LDS + merge by rank (nms_merge_kernel).  Quantised scores make
    the keep list depend on the tie order across chunk borders; batched form with the category offsets of the whole set."""
    from object_detectors_amd.tvision import boxes as B
    boxes, scores = nms_inputs(7000 + n, n, extent=6000.0)
    scores = (np.floor(scores * 64) / 64).astype(np.float32)
    keep = B.nms(T(boxes), T(scores), 0.5).cpu().numpy()
    assert np.array_equal(keep, tv.nms(boxes, scores, 0.5))
    idxs = detrand.randint(91 + n, (n,), 0, 7)
    keep = B.batched_nms(T(boxes), T(scores), T(idxs), 0.5).cpu().numpy()
    assert np.array_equal(keep, tv.batched_nms(boxes, scores, idxs, 0.5))


@pytest.mark.parametrize("k", [1, 5, 90, 1203])
def test_batched_nms_gpu(k):
    from object_detectors_amd.tvision import boxes as B
    boxes, scores = nms_inputs(900 + k, 5000)
    idxs = detrand.randint(77 + k, (5000,), 0, k)
    keep = B.batched_nms(T(boxes), T(scores), T(idxs), 0.5).cpu().numpy()
    assert np.array_equal(keep, tv.batched_nms(boxes, scores, idxs, 0.5))
    assert B.batched_nms(torch.zeros(0, 4, device=dev()), torch.zeros(0, device=dev()), torch.zeros(0, dtype=torch.int64, device=dev()), 0.5).numel() == 0


def test_sigmoid_focal_loss_gpu():
    from object_detectors_amd.tvision.focal_loss import sigmoid_focal_loss
    x = detrand.uniform(11, (4000, 91), -8, 8)
    t = (detrand.uniform(12, (4000, 91), 0, 1) > 0.98).astype(np.float32)
    xt = T(x).requires_grad_(True)
    loss = sigmoid_focal_loss(xt, T(t), reduction="sum")
    loss.backward()
    l, g = tv.sigmoid_focal_loss(x, t)
    np.testing.assert_allclose(loss.item(), l.astype(np.float64).sum(), rtol=1e-4)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g, rtol=1e-3, atol=1e-6)
    # reduction='none' (torchvision's default argument) and 'mean': elementwise loss with autograd, any shape; a bad mode raises like torchvision
    xn = T(x.reshape(40, 100, 91)).requires_grad_(True)
    wts = T(detrand.uniform(13, (40, 100, 91), 0.5, 1.5))
    ln = sigmoid_focal_loss(xn, T(t.reshape(40, 100, 91)))
    assert ln.shape == xn.shape
    np.testing.assert_allclose(ln.detach().cpu().numpy().reshape(4000, 91), l, rtol=1e-4, atol=1e-7)
    (ln * wts).sum().backward()
    np.testing.assert_allclose(xn.grad.cpu().numpy().reshape(4000, 91), g * wts.cpu().numpy().reshape(4000, 91), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(sigmoid_focal_loss(T(x), T(t), reduction="mean").item(), l.astype(np.float64).mean(), rtol=1e-4)
    with pytest.raises(ValueError):
        sigmoid_focal_loss(T(x), T(t), reduction="avg")
    e = sigmoid_focal_loss(torch.zeros(0, 5, device=dev()), torch.zeros(0, 5, device=dev()))
    assert e.shape == (0, 5)


def test_retinanet_cls_loss_gpu(golden):
    from object_detectors_amd.tvision._utils import Matcher
    from object_detectors_amd.tvision.focal_loss import retinanet_classification_loss
    g = golden("g5_7_tvision")
    anchors = build_anchors(g, "retina_small")
    N, K, b = anchors.shape[0], 91, 2
    gts = []
    for i in range(b):
        side = detrand.uniform(50 + i, (4, 2), 16, 90)
        tl = detrand.uniform(60 + i, (4, 2), 0, 1) * (np.array([160, 128], np.float32) - side)
        gts.append((np.concatenate([tl, tl + side], 1).astype(np.float32), detrand.randint(70 + i, (4,), 1, K)))
    logits = detrand.uniform(80, (b, N, K), -6, 2)
    reg = detrand.uniform(81, (b, N, 4), -1, 1)
    tfidf = detrand.uniform(82, (K,), 0.5, 2.0)
    for tf in (None, tfidf):
        cl, _rl, mis, (gc, _gr) = tv.retinanet_loss(logits, reg, anchors, gts, tfidf=tf)
        m = Matcher(0.5, 0.4, True)
        lt = T(logits).requires_grad_(True)
        matched = [m.match_boxes(T(bx), T(anchors)) for bx, _ in gts]
        for a, bm in zip(matched, mis):
            assert np.array_equal(a.cpu().numpy(), bm)
        targets = [{"labels": T(lb)} for _, lb in gts]
        loss = retinanet_classification_loss(lt, targets, matched, tfidf=None if tf is None else T(tf))
        loss.backward()
        np.testing.assert_allclose(loss.item(), cl, rtol=1e-4)
        np.testing.assert_allclose(lt.grad.cpu().numpy(), gc, rtol=2e-3, atol=1e-7)


def test_nms_batch_equals_per_image_nms():
    """mi355det_nms_batch: `bs` independent NMS problems side by side in one launch sequence == mi355det_nms per image (keep order and
    counts), with and without per-box categories; covers the bitonic (n < 4096), radix and chunked (n > 16384) sorts."""
    from object_detectors_amd import ops
    for n, bs, seed in ((300, 3, 1), (4750, 4, 2), (20000, 2, 3)):
        g = torch.Generator().manual_seed(seed)
        ctr = torch.rand((bs, n, 2), generator=g) * 400
        wh = torch.rand((bs, n, 2), generator=g) * 60 + 2
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).cuda()
        scores = torch.rand((bs, n), generator=g).cuda()
        scores[0, : n // 7] = float("-inf")                                   # masked entries rank last (rpn_filter_proposals)
        cats = torch.randint(0, 5, (bs, n), generator=g).cuda()
        for idxs in (None, cats):
            keep, cnt = ops.nms_batch(boxes, scores, 0.6, idxs=idxs)
            torch.cuda.synchronize()
            for b in range(bs):
                k1, c1 = ops.nms_raw(boxes[b], scores[b], 0.6, idxs=None if idxs is None else idxs[b])
                c = int(c1.item())
                assert int(cnt[b].item()) == c
                assert torch.equal(keep[b, :c], k1[:c])
