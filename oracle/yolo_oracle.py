"""CPU oracle for the YOLO side of the hot path — TEST INFRASTRUCTURE ONLY.

A float32 numpy restatement of the reference's algorithm, operation order preserved so that
index decisions (argmax, thresholds, NMS keeps) are bit-exact.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this; the product path never does.

Parity pinned: every function here is checked against tests/golden/*.npz, which
tools/make_golden.py produced by running the reference's own Python in the build container.

Reference files restated (all under /root/reference/yolo):
  utilities/helper.py:203-217   get_abs_coord
  utilities/helper.py:221-277   bbox_iou
  utilities/helper.py:280-382   nms_majority
  utilities/custom.py:40-67     FocalLoss
  nets/yolo_forw.py:81-251      YOLOForw.forward / get_target / transform_pred / get_stats
  procedures/test_one_epoch.py:24-37  post-processing
"""
import math

import numpy as np

F32 = np.float32
EPS = F32(1e-16)


def get_abs_coord(box):
    """helper.py:203-217 — xcycwh -> xyxy, last axis."""
    box = np.asarray(box, F32)
    x1 = box[..., 0] - box[..., 2] / F32(2)
    y1 = box[..., 1] - box[..., 3] / F32(2)
    x2 = box[..., 0] + box[..., 2] / F32(2)
    y2 = box[..., 1] + box[..., 3] / F32(2)
    return np.stack((x1, y1, x2, y2), axis=-1)


def bbox_iou(bb1, bb2, iou_type, xcycwh=True):
    """helper.py:221-277 — broadcast IoU(0)/GIoU(1)/DIoU(2)/CIoU(3), float32, same op order."""
    b1 = get_abs_coord(bb1) if xcycwh else np.asarray(bb1, F32)
    b2 = get_abs_coord(bb2) if xcycwh else np.asarray(bb2, F32)
    b1_x1, b1_y1, b1_x2, b1_y2 = b1[..., 0], b1[..., 1], b1[..., 2], b1[..., 3]
    b2_x1, b2_y1, b2_x2, b2_y2 = b2[..., 0], b2[..., 1], b2[..., 2], b2[..., 3]
    inter = np.maximum(np.minimum(b1_x2, b2_x2) - np.maximum(b1_x1, b2_x1), F32(0)) * \
        np.maximum(np.minimum(b1_y2, b2_y2) - np.maximum(b1_y1, b2_y1), F32(0))
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1
    union = (w1 * h1 + EPS) + w2 * h2 - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union
        if iou_type in (1, 2, 3):
            cw = np.maximum(b1_x2, b2_x2) - np.minimum(b1_x1, b2_x1)
            ch = np.maximum(b1_y2, b2_y2) - np.minimum(b1_y1, b2_y1)
            if iou_type == 1:
                c_area = cw * ch + EPS
                return iou - (c_area - union) / c_area
            c2 = cw * cw + ch * ch + EPS
            rho2 = ((b2_x1 + b2_x2) - (b1_x1 + b1_x2)) ** 2 / F32(4) + \
                   ((b2_y1 + b2_y2) - (b1_y1 + b1_y2)) ** 2 / F32(4)
            if iou_type == 2:
                return iou - rho2 / c2
            v = F32(4 / math.pi ** 2) * (np.arctan(w2 / h2) - np.arctan(w1 / h1)) ** 2
            alpha = v / (F32(1) - iou + v)
            return iou - (rho2 / c2 + v * alpha)
    return iou


def nms_majority(P, thresh_iou=0.6):
    """helper.py:280-382.  Returns (kept rows [k,6] with relabelled class, kept original indices).

    Stable ascending sort stands in for torch's argsort (equal scores: unspecified in the
    reference; the build defines 'stable by original index', SURVEY §7 iii).
    """
    P = np.array(P, F32, copy=True)
    x1, y1, x2, y2, scores = P[:, 0], P[:, 1], P[:, 2], P[:, 3], P[:, 4]
    classes = P[:, 5].astype(np.int32)          # snapshot: later relabels do not affect votes
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(scores, kind="stable")
    thr = F32(thresh_iou)
    keep = []
    while len(order) > 0:
        idx = order[-1]
        order = order[:-1]
        keep.append(idx)
        if len(order) == 0:
            break
        xx1 = np.maximum(x1[order], x1[idx])
        yy1 = np.maximum(y1[order], y1[idx])
        xx2 = np.minimum(x2[order], x2[idx])
        yy2 = np.minimum(y2[order], y2[idx])
        w = np.maximum(xx2 - xx1, F32(0))
        h = np.maximum(yy2 - yy1, F32(0))
        inter = w * h
        union = (areas[order] - inter) + areas[idx]
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / union
        mask = iou < thr
        sup = classes[order[iou > thr]]
        if sup.shape[0] > 0:
            cats, cnts = np.unique(sup, return_counts=True)
            if cnts.shape[0] > 1:
                P[idx, 5] = F32(cats[np.argmax(cnts)])   # first max -> smallest class id on ties
        order = order[mask]
    keep = np.array(keep, np.int64)
    return P[keep], keep


def focal_loss(pred, true, gamma, alpha):
    """custom.py:50-60 unreduced loss and its derivative wrt pred (float64 internally for the grad)."""
    x = np.asarray(pred, F32)
    t = np.asarray(true, F32)
    bce = np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))
    p = F32(1) / (F32(1) + np.exp(-x))
    p_t = t * p + (1 - t) * (1 - p)
    af = t * F32(alpha) + (1 - t) * F32(1 - alpha)
    mf = (F32(1) - p_t) ** F32(gamma)
    loss = (bce * (af * mf)).astype(F32)
    # derivative
    xd, td = x.astype(np.float64), t.astype(np.float64)
    pd = 1 / (1 + np.exp(-xd))
    ptd = td * pd + (1 - td) * (1 - pd)
    bced = np.maximum(xd, 0) - xd * td + np.log1p(np.exp(-np.abs(xd)))
    dbce = pd - td
    dpt = (2 * td - 1) * pd * (1 - pd)
    q = 1 - ptd
    with np.errstate(divide="ignore", invalid="ignore"):
        dmf = np.where(q > 0, -gamma * q ** (gamma - 1) * dpt, 0.0)
    grad = af.astype(np.float64) * (dbce * q ** gamma + bced * dmf)
    return loss, grad.astype(F32)


class YoloSpec:
    """Static configuration of the criterion (yolo_forw.py:13-77 with hydra/yolo/head.yaml defaults)."""

    def __init__(self, anchors, num_classes, img_size, iou_type=1, ignore_thr=0.5, lambda_iou=1.0,
                 lambda_xy=2.5, lambda_wh=2.5, lambda_conf=1.0, lambda_no_conf=0.1, lambda_cls=1.0,
                 alpha=0.5, gamma=1.0, idf_logits=None, class_weights=None, class_loss=1, reduction="sum", eq_mask=None):
        self.anchors = [[(float(w), float(h)) for w, h in s] for s in anchors]
        self.na = len(self.anchors[0])
        self.C = num_classes
        self.attrs = 5 + num_classes
        self.img_size = img_size
        self.iou_type = iou_type
        self.ignore_thr = ignore_thr
        self.l_iou, self.l_xy, self.l_wh = lambda_iou, lambda_xy, lambda_wh
        self.l_conf, self.l_noconf, self.l_cls = lambda_conf, lambda_no_conf, lambda_cls
        self.alpha, self.gamma = alpha, gamma
        self.idf = None if idf_logits is None else np.asarray(idf_logits, F32)
        self.cw = None if class_weights is None else np.asarray(class_weights, F32)      # CrossEntropyLoss(weight=...) / pos_weight (yolo_forw.py:50-62,70-77)
        self.class_loss = class_loss       # 0 BCEWithLogits(pos_weight) | 1 CrossEntropy(weight) | 2 EQLoss over BCE (yolo_forw.py:69-77)
        self.reduction = reduction         # 'sum' (then / nG, yolo_forw.py:158-160) | 'mean'
        self.eq_mask = None if eq_mask is None else np.asarray(eq_mask, F32)             # custom.py:79-80 (img_freq share < 0.0045)


def anchor_table(spec, grids):
    """yolo_forw.py:93-119 — cxypwh [N,4] (normalised) and inw_inh [N]; index (y*W+x)*A+a per scale."""
    cx, inw = [], []
    for k, g in enumerate(grids):
        stride = F32(spec.img_size / g)
        # torch.tensor([(a_w/stride_w, a_h/stride_h)]) — python float64 division, then float32
        sa = np.array([(aw / (spec.img_size / g), ah / (spec.img_size / g)) for aw, ah in spec.anchors[k]], F32)
        lin = np.linspace(0, g - 1, g, dtype=F32) + F32(0.5)
        gx = np.broadcast_to(lin[None, :, None], (g, g, spec.na)).reshape(-1) / F32(g)
        gy = np.broadcast_to(lin[:, None, None], (g, g, spec.na)).reshape(-1) / F32(g)
        aw = np.broadcast_to((sa[:, 0] / F32(g))[None, None, :], (g, g, spec.na)).reshape(-1)
        ah = np.broadcast_to((sa[:, 1] / F32(g))[None, None, :], (g, g, spec.na)).reshape(-1)
        cx.append(np.stack((gx, gy, aw, ah), axis=1))
        inw.append(np.full(g * g * spec.na, g, F32))
        del stride
    return np.concatenate(cx, 0).astype(F32), np.concatenate(inw, 0)


def flatten_heads(spec, heads):
    """yolo_forw.py:101-103,118 — NCHW heads -> raw_pred [bs, N, attrs]."""
    out = []
    for h in heads:
        bs, _, H, W = h.shape
        p = h.reshape(bs, spec.na, spec.attrs, H, W).transpose(0, 3, 4, 1, 2).reshape(bs, -1, spec.attrs)
        out.append(p)
    return np.concatenate(out, 1).astype(F32)


def get_target(spec, targets, cxypwh, inw_inh):
    """yolo_forw.py:178-208.  targets: list of (bbox [M,4] f32 rel xcycwh, labels [M] i64)."""
    obj_idx, noobj, tgt = [], [], []
    thr = F32(spec.ignore_thr)
    for bbox, _lab in targets:
        bbox = np.asarray(bbox, F32)
        iou = bbox_iou(bbox[:, None, :], cxypwh[None, :, :], spec.iou_type)
        best = np.argmax(iou, axis=1)                 # first index on ties (numpy == torch CPU)
        gt = cxypwh[best]
        in_wh = inw_inh[best]
        gx = bbox[:, 0] * in_wh - np.trunc(bbox[:, 0] * in_wh)
        gy = bbox[:, 1] * in_wh - np.trunc(bbox[:, 1] * in_wh)
        gx = np.clip(gx, F32(0.0001), F32(0.9999))
        gy = np.clip(gy, F32(0.0001), F32(0.9999))
        gw = np.log(bbox[:, 2] / gt[:, 2] + EPS)
        gh = np.log(bbox[:, 3] / gt[:, 3] + EPS)
        tgt.append(np.stack([gx, gy, gw, gh], 1).astype(F32))
        nm = np.all(iou < thr, axis=0)
        nm[best] = False
        noobj.append(nm)
        obj_idx.append(best.astype(np.int64))
    return np.concatenate(tgt, 0), obj_idx, np.stack(noobj, 0)


def _sigmoid(x):
    return F32(1) / (F32(1) + np.exp(-x.astype(F32)))


def decode_rows(spec, rows, cx, inw):
    """yolo_forw.py:166-167 / 217-218 on [...,4] raw values with matching anchor rows."""
    strides = F32(spec.img_size) / inw
    xy = (_sigmoid(rows[..., 0:2]) + cx[..., :2] * inw[..., None] - F32(0.5)) * strides[..., None]
    wh = np.exp(rows[..., 2:4]) * cx[..., 2:4] * inw[..., None] * strides[..., None]
    return np.concatenate([xy, wh], -1).astype(F32)


def decode(spec, heads):
    """Inference branch, yolo_forw.py:163-176 -> [bs, N, attrs]."""
    grids = [h.shape[2] for h in heads]
    cx, inw = anchor_table(spec, grids)
    raw = flatten_heads(spec, heads)
    box = decode_rows(spec, raw[..., :4], cx[None], np.broadcast_to(inw[None], raw.shape[:2]))
    conf = _sigmoid(raw[..., 4:5])
    logits = raw[..., 5:] if spec.idf is None else spec.idf[None, None, :] * raw[..., 5:]
    if spec.class_loss == 1:
        m = logits.max(-1, keepdims=True)
        e = np.exp(logits - m)
        cls = e / e.sum(-1, keepdims=True)
    else:
        cls = _sigmoid(logits)                                # yolo_forw.py:171-173
    return np.concatenate([box, conf, cls], -1).astype(F32)


def bce_class_loss(spec, logits, onehot):
    """class_loss 0 / 2 (yolo_forw.py:70-77): BCEWithLogitsLoss(pos_weight) elementwise, for 2 wrapped in custom.EQLoss (custom.py:83-99).
    -> (loss [nG,C] f32, d loss / d logits f64)."""
    x = logits.astype(np.float64)
    y = onehot.astype(np.float64)
    pw = np.ones(spec.C) if spec.cw is None else spec.cw.astype(np.float64)
    lw = 1 + (pw[None, :] - 1) * y
    sp = np.log1p(np.exp(-np.abs(x))) + np.maximum(-x, 0)               # softplus(-x)
    bce = (1 - y) * x + lw * sp
    sig = 1 / (1 + np.exp(-x))
    dbce = (1 - y) - lw * (1 - sig)
    if spec.class_loss == 0:
        return bce.astype(F32), dbce
    p_t = y * sig + (1 - y) * (1 - sig)
    af = y * spec.alpha + (1 - y) * (1 - spec.alpha)
    q = 1 - p_t
    w = np.clip(spec.eq_mask.astype(np.float64)[None, :] + y, 0.0, 1.0)
    mf = q ** spec.gamma
    dpt = (2 * y - 1) * sig * (1 - sig)
    with np.errstate(divide="ignore", invalid="ignore"):
        dmf = np.where(q > 0, -spec.gamma * q ** (spec.gamma - 1) * dpt, 0.0)
    return (bce * af * mf * w).astype(F32), w * af * (dbce * mf + bce * dmf)


def yolo_loss(spec, heads, targets, want_grad=True):
    """Train branch, yolo_forw.py:121-162.  Returns dict(loss, sub_losses[6], stats[5], tgt, obj_idx,
    noobj, grads per head).  Sums in float64 then rounded (reduction order is not part of parity)."""
    grids = [h.shape[2] for h in heads]
    cx, inw = anchor_table(spec, grids)
    raw = flatten_heads(spec, heads)
    bs, N, A = raw.shape
    tgt, obj_idx, noobj = get_target(spec, targets, cx, inw)
    nG = tgt.shape[0]
    bidx = np.concatenate([np.full(len(o), b) for b, o in enumerate(obj_idx)])
    aidx = np.concatenate(obj_idx)
    final = raw[bidx, aidx]                                  # [nG, attrs]
    labels = np.concatenate([np.asarray(l, np.int64) for _, l in targets])
    cxs, inws = cx[aidx], inw[aidx]
    pbox = decode_rows(spec, final[:, :4], cxs, inws)
    gbox = decode_rows_target(spec, tgt, cxs, inws)
    iou = bbox_iou(pbox, gbox, spec.iou_type)
    sxy = _sigmoid(final[:, :2])
    f64 = np.float64
    mean = spec.reduction != "sum"
    no_obj = raw[..., 4][noobj]
    onehot = np.zeros((nG, spec.C), bool)
    onehot[np.arange(nG), labels] = True
    # per-term divisors: 'sum' -> everything / nG at the end (yolo_forw.py:158-160); 'mean' -> each loss its own element count
    # (MSELoss over [nG,2]; FocalLoss / iou over nG; no-object term over its element count; class term below)
    d_xy = 2.0 * nG if mean else 1.0
    d_box = float(nG) if mean else 1.0
    d_no = float(max(no_obj.size, 1)) if mean else 1.0
    loss_xy = spec.l_xy * ((sxy - tgt[:, :2]).astype(f64) ** 2).sum() / d_xy
    loss_wh = spec.l_wh * ((final[:, 2:4] - tgt[:, 2:4]).astype(f64) ** 2).sum() / d_xy
    pl, pg = focal_loss(final[:, 4], np.ones(nG, F32), spec.gamma, spec.alpha)
    pos_conf = spec.l_conf * pl.astype(f64).sum() / d_box
    nl, ng = focal_loss(no_obj, np.zeros_like(no_obj), spec.gamma, spec.alpha)
    neg_conf = spec.l_noconf * nl.astype(f64).sum() / d_no
    logits = final[:, 5:] if spec.idf is None else spec.idf[None, :] * final[:, 5:]
    m = logits.max(-1, keepdims=True)
    wy = np.ones(nG, F32) if spec.cw is None else spec.cw[labels]
    if spec.class_loss == 1:
        lse = m[:, 0] + np.log(np.exp(logits - m).sum(-1))
        ce = lse - logits[np.arange(nG), labels]
        d_cls = float(wy.astype(f64).sum()) if mean else 1.0          # CrossEntropyLoss(weight, 'mean') divides by the summed target weights
        cls_loss = spec.l_cls * (wy.astype(f64) * ce.astype(f64)).sum() / d_cls
    else:
        bl, dbl = bce_class_loss(spec, logits, onehot)
        d_cls = float(nG * spec.C) if mean else 1.0
        cls_loss = spec.l_cls * bl.astype(f64).sum() / d_cls
    iou_loss = spec.l_iou * (1 - iou.astype(f64)).sum() / d_box
    sub = np.array([loss_xy, loss_wh, iou_loss, pos_conf, neg_conf, cls_loss], f64)
    final_div = 1.0 if mean else float(nG)
    loss = sub.sum() / final_div
    # stats (yolo_forw.py:233-248): true_pred classes = softmax (CE) / sigmoid (BCE forms) of RAW logits (no idf), transform_pred :220-223
    if spec.class_loss == 1:
        rm = final[:, 5:].max(-1, keepdims=True)
        pe = np.exp(final[:, 5:] - rm)
        pcls = pe / pe.sum(-1, keepdims=True)
    else:
        pcls = _sigmoid(final[:, 5:])
    stats = np.array([iou.astype(f64).mean(), _sigmoid(final[:, 4]).astype(f64).mean(),
                      _sigmoid(no_obj).astype(f64).mean(), pcls[onehot].astype(f64).mean(),
                      pcls[~onehot].astype(f64).mean()], f64)
    out = {"loss": F32(loss), "sub_losses": (sub / final_div).astype(F32), "stats": stats.astype(F32),
           "tgt": tgt, "obj_idx": obj_idx, "noobj": noobj}
    if want_grad:
        g = np.zeros((bs, N, A), f64)
        inv = 1.0 / final_div
        conf = raw[..., 4]
        _, gall = focal_loss(conf, np.zeros_like(conf), spec.gamma, spec.alpha)
        g[..., 4] = np.where(noobj, spec.l_noconf * gall.astype(f64) * inv / d_no, 0.0)
        giou = _iou_grad_fd(spec, final[:, :4], tgt, cxs, inws)
        e = np.exp(logits - m)
        sm = e / e.sum(-1, keepdims=True)
        for i in range(nG):                                  # duplicates accumulate
            b, a = bidx[i], aidx[i]
            s = sxy[i].astype(f64)
            g[b, a, 0:2] += spec.l_xy * 2 * (s - tgt[i, :2]) * s * (1 - s) * inv / d_xy
            g[b, a, 2:4] += spec.l_wh * 2 * (final[i, 2:4].astype(f64) - tgt[i, 2:4]) * inv / d_xy
            g[b, a, 0:4] += -spec.l_iou * giou[i] * inv / d_box
            g[b, a, 4] += spec.l_conf * pg[i] * inv / d_box
            if spec.class_loss == 1:
                dl = sm[i].astype(f64).copy()
                dl[labels[i]] -= 1
                dl = dl * float(wy[i])
            else:
                dl = dbl[i].copy()
            if spec.idf is not None:
                dl = dl * spec.idf
            g[b, a, 5:] += spec.l_cls * dl * inv / d_cls
        out["grad_flat"] = g.astype(F32)
        out["grads"] = unflatten_grads(spec, g.astype(F32), heads)
    return out


def decode_rows_target(spec, tgt, cx, inw):
    """yolo_forw.py:227-229 — target box in pixels (no sigmoid on xy)."""
    strides = F32(spec.img_size) / inw
    xy = (tgt[:, :2] + cx[:, :2] * inw[:, None] - F32(0.5)) * strides[:, None]
    wh = np.exp(tgt[:, 2:4]) * cx[:, 2:4] * inw[:, None] * strides[:, None]
    return np.concatenate([xy, wh], 1).astype(F32)


def _iou_grad_fd(spec, raw4, tgt, cxs, inws):
    """d IoU-metric / d raw[0:4] by float64 central differences of the same formulae (oracle only)."""
    def f(r):
        r = r.astype(np.float64)
        st = spec.img_size / inws.astype(np.float64)
        sx = 1 / (1 + np.exp(-r[:, :2]))
        xy = (sx + cxs[:, :2].astype(np.float64) * inws[:, None] - 0.5) * st[:, None]
        wh = np.exp(r[:, 2:4]) * cxs[:, 2:4].astype(np.float64) * inws[:, None] * st[:, None]
        p = np.concatenate([xy, wh], 1)
        g = decode_rows_target(spec, tgt, cxs, inws).astype(np.float64)
        return _iou64(p, g, spec.iou_type, alpha0)
    alpha0 = None
    alpha0 = f(raw4)[1] if spec.iou_type == 3 else None
    f_ = f
    f = lambda r: (f_(r)[0] if spec.iou_type == 3 else f_(r))  # noqa: E731
    out = np.zeros((raw4.shape[0], 4))
    h = 1e-5
    for k in range(4):
        d = np.zeros(4)
        d[k] = h
        out[:, k] = (f(raw4.astype(np.float64) + d) - f(raw4.astype(np.float64) - d)) / (2 * h)
    return out


def _iou64(b1, b2, iou_type, alpha_const=None):
    x1a, y1a, x2a, y2a = b1[:, 0] - b1[:, 2] / 2, b1[:, 1] - b1[:, 3] / 2, b1[:, 0] + b1[:, 2] / 2, b1[:, 1] + b1[:, 3] / 2
    x1b, y1b, x2b, y2b = b2[:, 0] - b2[:, 2] / 2, b2[:, 1] - b2[:, 3] / 2, b2[:, 0] + b2[:, 2] / 2, b2[:, 1] + b2[:, 3] / 2
    inter = np.clip(np.minimum(x2a, x2b) - np.maximum(x1a, x1b), 0, None) * \
        np.clip(np.minimum(y2a, y2b) - np.maximum(y1a, y1b), 0, None)
    w1, h1, w2, h2 = x2a - x1a, y2a - y1a, x2b - x1b, y2b - y1b
    union = (w1 * h1 + 1e-16) + w2 * h2 - inter
    iou = inter / union
    if iou_type == 0:
        return iou
    cw = np.maximum(x2a, x2b) - np.minimum(x1a, x1b)
    ch = np.maximum(y2a, y2b) - np.minimum(y1a, y1b)
    if iou_type == 1:
        ca = cw * ch + 1e-16
        return iou - (ca - union) / ca
    c2 = cw ** 2 + ch ** 2 + 1e-16
    rho2 = ((x1b + x2b) - (x1a + x2a)) ** 2 / 4 + ((y1b + y2b) - (y1a + y2a)) ** 2 / 4
    if iou_type == 2:
        return iou - rho2 / c2
    v = (4 / math.pi ** 2) * (np.arctan(w2 / h2) - np.arctan(w1 / h1)) ** 2
    alpha = v / (1 - iou + v)          # treated as a constant by the reference (no_grad, helper.py:273)
    if alpha_const is not None:
        alpha = alpha_const
    return iou - (rho2 / c2 + v * alpha), alpha


def unflatten_grads(spec, g, heads):
    out, off = [], 0
    for h in heads:
        bs, _, H, W = h.shape
        n = H * W * spec.na
        gh = g[:, off:off + n].reshape(bs, H, W, spec.na, spec.attrs).transpose(0, 3, 4, 1, 2)
        out.append(np.ascontiguousarray(gh.reshape(bs, spec.na * spec.attrs, H, W)))
        off += n
    return out


def postprocess(pred, conf_thr=0.1, nms_thr=0.6):
    """test_one_epoch.py:24-37 on decoded predictions [bs,N,attrs] -> list of (cand [n,6], final [k,6])."""
    pred = np.array(pred, F32, copy=True)
    pred[..., :4] = get_abs_coord(pred[..., :4])
    cmax = pred[..., 5:].max(-1)
    carg = pred[..., 5:].argmax(-1)
    score = pred[..., 4] * cmax
    out = []
    for b in range(pred.shape[0]):
        m = score[b] > F32(conf_thr)
        if not m.any():
            continue
        cand = np.concatenate([pred[b][m][:, :4], score[b][m][:, None], carg[b][m][:, None].astype(F32)], 1)
        fin, _ = nms_majority(cand, nms_thr)
        out.append((cand, fin))
    return out


def minibatch_idf(label_lists, num_classes, norm=0):
    """IDFTransformer.forward (yolo/utilities/custom.py:257-262) + the optional p-normalisation of yolo_forw.py:87-91."""
    t = np.stack([np.bincount(np.asarray(l, np.int64), minlength=num_classes) for l in label_lists])
    df = (t > 0).sum(0).astype(np.float64)
    w = np.log((len(label_lists) + 1) / (df + 1)) + 1
    if norm != 0:
        w = w / np.linalg.norm(w, ord=norm)
    return w.astype(F32)
