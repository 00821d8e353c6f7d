"""CPU oracle for the torchvision_models side of the hot path — TEST INFRASTRUCTURE ONLY.

float32 numpy restatement.  Matcher / BoxCoder / AnchorGenerator follow reference files and are
pinned by tests/golden/g5_7_tvision.npz (reference outputs).  box_iou / nms / batched_nms /
clip_boxes_to_image / remove_small_boxes / sigmoid_focal_loss live in torchvision, which is
NOT vendored in /root/reference and not installed here (torchvision ~0.10, unpinned by the
reference): they restate the published semantics and are
**parity unpinned** beyond the reference-owned callers that consume them (Matcher on box_iou).

Reference files restated (under /root/reference/torchvision_models/tvision):
  _utils.py:79-125,152-223   BoxCoder.encode_single / decode_single
  _utils.py:271-344          Matcher.__call__ / set_low_quality_matches_
  anchor_utils.py:60-159     AnchorGenerator
  retinanet.py:107-143,196-223,401-412   RetinaNet losses
"""
import math

import numpy as np

F32 = np.float32


def box_iou(a, b):
    a, b = np.asarray(a, F32), np.asarray(b, F32)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.maximum(rb - lt, F32(0))
    inter = wh[..., 0] * wh[..., 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area_a[:, None] + area_b[None, :] - inter)


def nms(boxes, scores, thr):
    """Greedy NMS, suppress IoU > thr, kept indices by descending score (ties: lower index first)."""
    boxes, scores = np.asarray(boxes, F32), np.asarray(scores, F32)
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    alive = np.ones(len(order), bool)
    keep = []
    for i, idx in enumerate(order):
        if not alive[i]:
            continue
        keep.append(idx)
        rest = order[i + 1:]
        lt = np.maximum(boxes[rest, :2], boxes[idx, :2])
        rb = np.minimum(boxes[rest, 2:], boxes[idx, 2:])
        wh = np.maximum(rb - lt, F32(0))
        inter = wh[:, 0] * wh[:, 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[idx] + area[rest] - inter)
        alive[i + 1:] &= ~(iou > F32(thr))
    return np.array(keep, np.int64)


def batched_nms(boxes, scores, idxs, thr):
    boxes = np.asarray(boxes, F32)
    if boxes.shape[0] == 0:
        return np.zeros(0, np.int64)
    off = np.asarray(idxs).astype(F32) * (boxes.max() + F32(1))
    return nms(boxes + off[:, None], scores, thr)


def clip_boxes_to_image(boxes, size):
    h, w = size
    b = np.array(boxes, F32, copy=True)
    b[..., 0::2] = np.clip(b[..., 0::2], 0, F32(w))
    b[..., 1::2] = np.clip(b[..., 1::2], 0, F32(h))
    return b


def remove_small_boxes(boxes, min_size):
    b = np.asarray(boxes, F32)
    ws, hs = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    return np.nonzero((ws >= F32(min_size)) & (hs >= F32(min_size)))[0].astype(np.int64)


def sigmoid_focal_loss(x, t, alpha=0.25, gamma=2.0):
    """Unreduced loss and d/dx (float64 internally)."""
    x = np.asarray(x, np.float64)
    t = np.asarray(t, np.float64)
    p = 1 / (1 + np.exp(-x))
    ce = np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))
    p_t = p * t + (1 - p) * (1 - t)
    q = 1 - p_t
    loss = ce * q ** gamma
    dpt = (2 * t - 1) * p * (1 - p)
    grad = (p - t) * q ** gamma - ce * gamma * q ** (gamma - 1) * dpt
    if alpha >= 0:
        a_t = alpha * t + (1 - alpha) * (1 - t)
        loss, grad = a_t * loss, a_t * grad
    return loss.astype(F32), grad.astype(F32)


def matcher(q, high, low, allow_low_quality):
    """_utils.py:271-344 on a materialised [M,N] quality matrix -> int64 [N]."""
    q = np.asarray(q, F32)
    if q.size == 0:
        raise ValueError("No ground-truth boxes available for one of the images during training"
                         if q.shape[0] == 0 else
                         "No proposal boxes available for one of the images during training")
    vals = q.max(axis=0)
    matches = q.argmax(axis=0).astype(np.int64)
    allm = matches.copy()
    matches[vals < F32(low)] = -1
    matches[(vals >= F32(low)) & (vals < F32(high))] = -2
    if allow_low_quality:
        best = q.max(axis=1)
        _, pred = np.nonzero(q == best[:, None])
        matches[pred] = allm[pred]
    return matches


def encode_boxes(ref, prop, weights):
    ref, prop = np.asarray(ref, F32), np.asarray(prop, F32)
    wx, wy, ww, wh = [F32(w) for w in weights]
    ew, eh = prop[:, 2] - prop[:, 0], prop[:, 3] - prop[:, 1]
    ecx, ecy = prop[:, 0] + F32(0.5) * ew, prop[:, 1] + F32(0.5) * eh
    gw, gh = ref[:, 2] - ref[:, 0], ref[:, 3] - ref[:, 1]
    gcx, gcy = ref[:, 0] + F32(0.5) * gw, ref[:, 1] + F32(0.5) * gh
    return np.stack([wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh,
                     ww * np.log(gw / ew), wh * np.log(gh / eh)], 1).astype(F32)


def decode_boxes(codes, boxes, weights, clip=math.log(1000.0 / 16)):
    codes, boxes = np.asarray(codes, F32), np.asarray(boxes, F32)
    wx, wy, ww, wh = [F32(w) for w in weights]
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + F32(0.5) * w, boxes[:, 1] + F32(0.5) * h
    dx, dy = codes[:, 0::4] / wx, codes[:, 1::4] / wy
    dw = np.minimum(codes[:, 2::4] / ww, F32(clip))
    dh = np.minimum(codes[:, 3::4] / wh, F32(clip))
    pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
    pw, ph = np.exp(dw) * w[:, None], np.exp(dh) * h[:, None]
    out = np.stack([pcx - F32(0.5) * pw, pcy - F32(0.5) * ph, pcx + F32(0.5) * pw, pcy + F32(0.5) * ph], 2)
    return out.reshape(codes.shape[0], -1).astype(F32)


def cell_anchors(sizes, ratios):
    """anchor_utils.py:60-71 (float32 sqrt / reciprocal, round-half-even like torch.round)."""
    s = np.asarray(sizes, F32)
    r = np.asarray(ratios, F32)
    hr = np.sqrt(r)
    wr = F32(1) / hr
    ws = (wr[:, None] * s[None, :]).reshape(-1)
    hs = (hr[:, None] * s[None, :]).reshape(-1)
    return np.round(np.stack([-ws, -hs, ws, hs], 1) / F32(2)).astype(F32)


def anchors(sizes, ratios, image_size, grids):
    """anchor_utils.py:98-159 -> [sum(H*W*A), 4] xyxy; stride = image // grid."""
    out = []
    for sz, ar, (gh, gw) in zip(sizes, ratios, grids):
        base = cell_anchors(sz, ar)
        sh, sw = image_size[0] // gh, image_size[1] // gw
        sx = np.arange(gw, dtype=F32) * F32(sw)
        sy = np.arange(gh, dtype=F32) * F32(sh)
        yy, xx = np.meshgrid(sy, sx, indexing="ij")
        shifts = np.stack([xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)], 1)
        out.append((shifts[:, None, :] + base[None, :, :]).reshape(-1, 4))
    return np.concatenate(out, 0).astype(F32)


def retinanet_loss(cls_logits, bbox_reg, anchors_xyxy, gts, tfidf=None, high=0.5, low=0.4):
    """retinanet.py:401-412 + :107-143 + :196-223.  cls_logits [b,N,K], bbox_reg [b,N,4];
    gts list of (boxes [M,4] xyxy, labels [M]).  Returns (cls_loss, reg_loss, matched_idxs, grads)."""
    b, N, K = cls_logits.shape
    cl, rl, mis = [], [], []
    gcls = np.zeros(cls_logits.shape, np.float64)
    greg = np.zeros(bbox_reg.shape, np.float64)
    for i in range(b):
        boxes, labels = gts[i]
        if np.asarray(boxes).size == 0:
            mi = np.full(N, -1, np.int64)
        else:
            mi = matcher(box_iou(boxes, anchors_xyxy), high, low, True)
        mis.append(mi)
        fg = mi >= 0
        nfg = max(1, int(fg.sum()))
        tgt = np.zeros((N, K), F32)
        tgt[fg, np.asarray(labels)[mi[fg]]] = 1
        valid = mi != -2
        x = cls_logits[i] if tfidf is None else np.asarray(tfidf, F32)[None, :] * cls_logits[i]
        l, g = sigmoid_focal_loss(x[valid], tgt[valid])
        cl.append(l.astype(np.float64).sum() / nfg)
        gi = np.zeros((N, K))
        gi[valid] = g / nfg
        if tfidf is not None:
            gi = gi * np.asarray(tfidf, np.float64)[None, :]
        gcls[i] = gi / b
        idx = np.nonzero(fg)[0]
        tr = encode_boxes(np.asarray(boxes, F32)[mi[idx]], anchors_xyxy[idx], (1, 1, 1, 1)) if len(idx) else np.zeros((0, 4), F32)
        d = bbox_reg[i][idx].astype(np.float64) - tr
        rl.append(np.abs(d).sum() / nfg)
        greg[i][idx] = np.sign(d) / nfg / max(1, b)
    return F32(sum(cl) / b), F32(sum(rl) / max(1, b)), mis, (gcls.astype(F32), greg.astype(F32))


def _bilinear(f, y, x):
    H, W = f.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return 0.0
    y, x = max(y, 0.0), max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1 - ly, 1 - lx
    return hy * hx * f[yl, xl] + hy * lx * f[yl, xh] + ly * hx * f[yh, xl] + ly * lx * f[yh, xh]


def roi_align(feat, rois, out_size, scale, sampling_ratio=2, aligned=False):
    """torchvision.ops.roi_align, float64 loops — small cases only.  feat [N,C,H,W], rois [K,5]."""
    ph, pw = out_size
    K, C = rois.shape[0], feat.shape[1]
    out = np.zeros((K, C, ph, pw), np.float64)
    off = 0.5 if aligned else 0.0
    for k in range(K):
        b = int(rois[k, 0])
        x1, y1, x2, y2 = [float(v) * scale - off for v in rois[k, 1:]]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph, rw / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        for c in range(C):
            f = feat[b, c].astype(np.float64)
            for py in range(ph):
                for px in range(pw):
                    acc = 0.0
                    for iy in range(gh):
                        y = y1 + py * bh + (iy + 0.5) * bh / gh
                        for ix in range(gw):
                            x = x1 + px * bw + (ix + 0.5) * bw / gw
                            acc += _bilinear(f, y, x)
                    out[k, c, py, px] = acc / max(gh * gw, 1)
    return out.astype(F32)


def map_levels(boxes, k_min, k_max):
    s = np.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).astype(F32)
    k = np.floor(F32(4) + np.log2(s / F32(224)) + F32(1e-6))
    return (np.clip(k, k_min, k_max) - k_min).astype(np.int64)


# ---- Faster R-CNN target / loss side (tvision/rpn.py:179-213,282-318; tvision/roi_heads.py:22-98,627-651) ----------------
def rpn_assign(anchors_xyxy, gt, high=0.7, low=0.3):
    """-> (labels float32 {1, 0, -1}, matched_gt_boxes [N,4])."""
    if np.asarray(gt).size == 0:
        return np.zeros(len(anchors_xyxy), F32), np.zeros((len(anchors_xyxy), 4), F32)
    m = matcher(box_iou(gt, anchors_xyxy), high, low, True)
    lab = (m >= 0).astype(F32)
    lab[m == -1] = 0.0
    lab[m == -2] = -1.0
    return lab, np.asarray(gt, F32)[np.clip(m, 0, None)]


def smooth_l1_sum(x, t, beta):
    n = np.abs(x.astype(np.float64) - t.astype(np.float64))
    return np.where(n < beta, 0.5 * n * n / beta, n - 0.5 * beta).sum()


def rpn_loss(objectness, deltas, labels, reg_targets, pos_idx, neg_idx):
    """compute_loss with the sampler's outcome (pos_idx / neg_idx into the concatenated anchors) given."""
    sampled = np.concatenate([pos_idx, neg_idx])
    box = smooth_l1_sum(deltas[pos_idx], reg_targets[pos_idx], 1.0 / 9) / sampled.size
    x = objectness.reshape(-1)[sampled].astype(np.float64)
    y = labels[sampled].astype(np.float64)
    bce = (np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))).mean()
    return F32(bce), F32(box)


def roi_assign(proposals, gt_boxes, gt_labels, high=0.5, low=0.5):
    m = matcher(box_iou(gt_boxes, proposals), high, low, False)
    lab = np.asarray(gt_labels)[np.clip(m, 0, None)].astype(np.int64)
    lab[m == -1] = 0
    lab[m == -2] = -1
    return np.clip(m, 0, None), lab


def fastrcnn_loss(class_logits, box_regression, labels, reg_targets, loss_type="ce", weights=None, want_grad=False):
    """roi_heads.py:24-96 on logits that already carry the tf-idf row (the caller multiplies, as roi_heads.py:826-827 does).
    float64 inside.  -> (cls_loss, box_loss) or, with want_grad, (cls_loss, box_loss, d cls / d logits, d box / d box_regression)."""
    x = class_logits.astype(np.float64)
    n, k = x.shape
    rows = np.arange(n)
    if loss_type == "ce":
        mx = x.max(1, keepdims=True)
        se = np.exp(x - mx).sum(1)
        lse = mx[:, 0] + np.log(se)
        w = np.ones(n) if weights is None else np.asarray(weights, np.float64)[labels]
        cls = (w * (lse - x[rows, labels])).sum() / w.sum()                        # F.cross_entropy(weight=..., reduction='mean')
        g = np.exp(x - mx) / se[:, None]
        g[rows, labels] -= 1.0
        g *= (w / w.sum())[:, None]
    else:
        y = np.zeros_like(x)
        y[rows, labels] = 1.0
        y[:, 0] = 0.0
        p = 1.0 / (1.0 + np.exp(-x))
        bce = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
        if loss_type == "bce":
            cls = bce.sum() / n
            g = (p - y) / n
        elif loss_type == "focal_loss":
            pt = p * y + (1 - p) * (1 - y)
            at = 0.25 * y + 0.75 * (1 - y)
            cls = (at * bce * (1 - pt) ** 2).sum() / n
            dpt = (2 * y - 1) * p * (1 - p)
            g = at * ((1 - pt) ** 2 * (p - y) - 2 * (1 - pt) * dpt * bce) / n
        elif loss_type.startswith("gombit"):
            u = x - 1.96                                                            # :59-61
            c = np.clip(u, -3, 5)
            passg = (u >= -3) & (u <= 5)
            e = np.exp(-c)
            pe = np.exp(-e)                                                         # 1 / exp(exp(-c))
            b = np.where(y > 0.5, e, -np.log(1 - pe))                               # F.binary_cross_entropy(pestim, y)
            db = np.where(y > 0.5, -1 / pe, 1 / (1 - pe))
            dpe = pe * e
            if loss_type.endswith("fl"):
                pt = pe * y + (1 - pe) * (1 - y)
                at = 0.25 * y + 0.75 * (1 - y)
                cls = (at * b * (1 - pt) ** 2).sum() / n
                g = at * (2 * (1 - pt) * (-(2 * y - 1)) * b + (1 - pt) ** 2 * db) * dpe / n
            else:
                cls = b.sum() / n
                g = db * dpe / n
                if cls > 5:                                                         # :71-72
                    cls, g = cls / 4, g / 4
            g = np.where(passg, g, 0.0)
        else:
            raise ValueError(loss_type)
    pos = np.nonzero(labels > 0)[0]
    br = box_regression.reshape(n, -1, 4)
    box = smooth_l1_sum(br[pos, labels[pos]], reg_targets[pos], 1.0) / labels.size
    if not want_grad:
        return F32(cls), F32(box)
    gb = np.zeros(br.shape, np.float64)
    d = br[pos, labels[pos]].astype(np.float64) - reg_targets[pos].astype(np.float64)
    gb[pos, labels[pos]] = np.where(np.abs(d) < 1.0, d, np.sign(d)) / labels.size
    return F32(cls), F32(box), g, gb.reshape(n, -1)


def minibatch_tfidf(label_lists, num_classes, norm=0):
    """roi_heads.py:801-809."""
    w = np.stack([np.bincount(np.asarray(l), minlength=num_classes) for l in label_lists])
    w = (w > 0).sum(0).astype(np.float64)
    w = np.log((len(label_lists) + 1) / (w + 1)) + 1
    if norm != 0:
        w = w / np.linalg.norm(w, ord=norm)
    return w.astype(F32)


def roi_postprocess_detections(class_logits, box_regression, proposals, image_shapes, tfidf_post=1.0, loss_type="ce", score_thresh=0.05,
                               nms_thresh=0.5, detections_per_img=100, weights=(10.0, 10.0, 5.0, 5.0)):
    """RoIHeads.postprocess_detections (roi_heads.py:715-781) -> list of (boxes [d,4], scores [d], labels [d]) per image."""
    x = np.asarray(tfidf_post, F32) * class_logits.astype(F32)
    if loss_type == "ce":
        z = x - x.max(-1, keepdims=True)
        ez = np.exp(z)
        scores = (ez / ez.sum(-1, keepdims=True)).astype(F32)
    elif loss_type.startswith("gombit"):
        scores = (1 / (np.exp(np.exp(-np.asarray(tfidf_post, F32) * (class_logits.astype(F32) - F32(1.96)))))).astype(F32)
    else:
        scores = (1 / (1 + np.exp(-x))).astype(F32)
    k = class_logits.shape[-1]
    out, off = [], 0
    for props, shape in zip(proposals, image_shapes):
        r = len(props)
        codes = box_regression[off:off + r].reshape(r * k, 4)
        boxes = decode_boxes(codes, np.repeat(np.asarray(props, F32), k, 0), weights).reshape(r, k, 4)
        sc = scores[off:off + r]
        off += r
        boxes = clip_boxes_to_image(boxes, shape)
        labels = np.broadcast_to(np.arange(k)[None, :], sc.shape)
        boxes, sc, labels = boxes[:, 1:].reshape(-1, 4), sc[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)
        inds = np.nonzero(sc > F32(score_thresh))[0]
        boxes, sc, labels = boxes[inds], sc[inds], labels[inds]
        keep = remove_small_boxes(boxes, 1e-2)
        boxes, sc, labels = boxes[keep], sc[keep], labels[keep]
        keep = batched_nms(boxes, sc, labels, nms_thresh)[:detections_per_img]
        out.append((boxes[keep], sc[keep], labels[keep].astype(np.int64)))
    return out


# ---- input-side transform (tvision/transform.py:26-52,88-226,279-293; ATen upsample_bilinear2d, align_corners=False) ---------------------
def resized_size(h, w, min_size, max_size):
    # `self_min_size / min_size` with a Python float on the left and a float32 tensor on the right is Tensor.__rtruediv__: reciprocal() * scalar,
    # i.e. TWO float32 roundings (1/480*800 = 1.6666667, not 1.6666666): this decides 800 vs 799 rows
    mn, mx = F32(min(h, w)), F32(max(h, w))
    scale = float(min((F32(1) / mn) * F32(min_size), (F32(1) / mx) * F32(max_size)))
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))


def bilinear_resize(img, oh, ow):
    """img [..., h, w] float32 -> [..., oh, ow]; F.interpolate(mode='bilinear', align_corners=False) with the scale taken from the sizes."""
    img = np.asarray(img, F32)
    h, w = img.shape[-2:]

    def axis(n_in, n_out):
        r = F32(n_in) / F32(n_out)
        src = np.maximum(r * (np.arange(n_out, dtype=F32) + F32(0.5)) - F32(0.5), F32(0))
        i0 = src.astype(np.int64)
        i1 = i0 + (i0 < n_in - 1)
        l1 = (src - i0.astype(F32)).astype(F32)
        return i0, i1, (F32(1) - l1).astype(F32), l1
    y0, y1, h0, h1 = axis(h, oh)
    x0, x1, w0, w1 = axis(w, ow)
    top = w0 * img[..., y0, :][..., x0] + w1 * img[..., y0, :][..., x1]
    bot = w0 * img[..., y1, :][..., x0] + w1 * img[..., y1, :][..., x1]
    return (h0[:, None] * top + h1[:, None] * bot).astype(F32)


def rcnn_transform(images, min_size=800, max_size=1333, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), size_divisible=32, boxes=None):
    """GeneralizedRCNNTransform.forward in eval mode -> (batch [N,3,Hp,Wp], image_sizes, resized boxes)."""
    m, s = np.asarray(mean, F32)[:, None, None], np.asarray(std, F32)[:, None, None]
    outs, sizes, nb = [], [], []
    for i, img in enumerate(images):
        img = np.asarray(img, F32)
        h, w = img.shape[-2:]
        oh, ow = resized_size(h, w, min_size, max_size)
        outs.append(bilinear_resize((img - m) / s, oh, ow))
        sizes.append((oh, ow))
        if boxes is not None:
            nb.append(resize_boxes(boxes[i], (h, w), (oh, ow)))
    ph = int(math.ceil(max(x[0] for x in sizes) / float(size_divisible)) * size_divisible)
    pw = int(math.ceil(max(x[1] for x in sizes) / float(size_divisible)) * size_divisible)
    batch = np.zeros((len(images), outs[0].shape[0], ph, pw), F32)
    for i, o in enumerate(outs):
        batch[i, :, :o.shape[1], :o.shape[2]] = o
    return batch, sizes, nb


def resize_boxes(boxes, original_size, new_size):
    rh, rw = F32(new_size[0]) / F32(original_size[0]), F32(new_size[1]) / F32(original_size[1])
    b = np.asarray(boxes, F32)
    return np.stack([b[:, 0] * rw, b[:, 1] * rh, b[:, 2] * rw, b[:, 3] * rh], 1).astype(F32)
