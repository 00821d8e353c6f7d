"""Tensor-level wrappers over the C ABI (include/mi355det.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation below runs
in libmi355det.so.  All tensors must live on the GPU; nothing falls back to torch ops.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import HeadView, YoloGeom, YoloLossCfg, check, ptr, stream_ptr

# ---- storage format seen by the convolution helpers below: "bf16" (default) or, inside `with ops.storage("fp16"):`, the fp16 twins of the
#      entry points (include/mi355det_f16.h) and torch.float16 buffers.  The engines carry their own format (YoloV3Engine(storage=...)).
_STORAGE = ["bf16"]


def lib():
    return _lib.storage_lib(_STORAGE[0])


def act_dtype():
    return torch.float16 if _STORAGE[0] == "fp16" else torch.bfloat16


class storage:
    def __init__(self, fmt):
        if fmt not in ("bf16", "fp16"):
            raise ValueError("storage must be 'bf16' or 'fp16'")
        self.fmt = fmt

    def __enter__(self):
        self.prev = _STORAGE[0]
        _STORAGE[0] = self.fmt
        return self

    def __exit__(self, *exc):
        _STORAGE[0] = self.prev
        return False



def _f32c(t):
    if not t.is_cuda:
        raise ValueError("mi355det ops need CUDA/HIP tensors (no CPU fallback)")
    return t.contiguous().float()


# ----------------------------------------------------------------------------------------- YOLO
def make_geom(anchors, num_classes, img_size, grids, iou_type=1, ignore_thr=0.5):
    """anchors: per scale list of (w,h) pixels, scale order = head order (stride 32,16,8).
    Normalised anchor sizes are computed exactly like yolo_forw.py:99,108-113 (float64 divide,
    float32 store, float32 divide by grid)."""
    import numpy as np
    g = YoloGeom()
    ns, na = len(grids), len(anchors[0])
    if ns > _lib.MAX_SCALES or na > _lib.MAX_ANCHORS:
        raise ValueError("unsupported number of scales / anchors per scale")
    g.num_scales, g.na, g.num_classes = ns, na, num_classes
    g.img_size, g.ignore_thr, g.iou_type = float(img_size), float(ignore_thr), int(iou_type)
    off = 0
    for k, gr in enumerate(grids):
        g.grid[k] = int(gr)
        g.off[k] = off
        off += gr * gr * na
        stride = img_size / gr
        for a, (aw, ah) in enumerate(anchors[k]):
            g.anchor_w[k][a] = float(np.float32(np.float32(aw / stride) / np.float32(gr)))
            g.anchor_h[k][a] = float(np.float32(np.float32(ah / stride) / np.float32(gr)))
    for k in range(ns, _lib.MAX_SCALES + 1):
        g.off[k] = off
    return g


def head_views(heads, attrs_total, dtype=torch.float32):
    """NCHW-shaped tensors [bs, A*attrs, H, W] with any (dense-pixel) strides -> HeadView array."""
    arr = (HeadView * _lib.MAX_SCALES)()
    keep = []
    for k, t in enumerate(heads):
        if t.dim() != 4 or t.shape[1] != attrs_total:
            raise ValueError(f"head {k}: expected [bs,{attrs_total},H,W], got {tuple(t.shape)}")
        if t.dtype != dtype or t.stride(2) != t.shape[3] * t.stride(3):
            if dtype != torch.float32:
                raise ValueError("gradient views must already be bf16 with dense pixel strides")
            t = t.float().contiguous()
        keep.append(t)
        arr[k].ptr = t.data_ptr()
        arr[k].sb, arr[k].sc, arr[k].sp = t.stride(0), t.stride(1), t.stride(3)
    return arr, keep


def flatten_targets(targets, device):
    """list of {'bbox': [M,4], 'category_id': [M]} -> (boxes [G,4] f32, labels [G] i64, off [bs+1] i32, counts)."""
    counts = [int(t["bbox"].shape[0]) for t in targets]
    boxes = torch.cat([t["bbox"].reshape(-1, 4) for t in targets]).to(device=device, dtype=torch.float32).contiguous()
    labels = torch.cat([t["category_id"].reshape(-1) for t in targets]).to(device=device, dtype=torch.int64).contiguous()
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    off = torch.tensor(offs, dtype=torch.int32).to(device, non_blocking=True)
    return boxes, labels, off, counts


def bbox_iou(bb1, bb2, iou_type, xcycwh=True):
    """helper.bbox_iou: broadcast [M,1,4]x[1,N,4] -> [M,N], or same-shape [n,4] -> [n]."""
    bb1, bb2 = _f32c(bb1), _f32c(bb2)
    if bb1.dim() == 3 and bb2.dim() == 3 and bb1.shape[1] == 1 and bb2.shape[0] == 1:
        m, n = bb1.shape[0], bb2.shape[1]
        out = torch.empty((m, n), device=bb1.device, dtype=torch.float32)
        check(lib().mi355det_bbox_iou(ptr(bb1), ptr(bb2), ptr(out), m, n, int(iou_type), int(xcycwh), 0, stream_ptr()), "bbox_iou")
        return out
    if bb1.shape != bb2.shape or bb1.shape[-1] != 4:
        raise ValueError("bbox_iou: shapes must be [M,1,4]x[1,N,4] or equal [...,4]")
    n = bb1.numel() // 4
    out = torch.empty(bb1.shape[:-1], device=bb1.device, dtype=torch.float32)
    check(lib().mi355det_bbox_iou(ptr(bb1), ptr(bb2), ptr(out), 1, n, int(iou_type), int(xcycwh), 1, stream_ptr()), "bbox_iou")
    return out


def yolo_assign(geom, boxes, off, bs, counts):
    """YOLOForw.get_target for the whole batch -> (obj_idx [G] i64, tgt [G,4], noobj [bs,N] u8)."""
    dev = boxes.device
    G, N = boxes.shape[0], geom.off[geom.num_scales]
    key = torch.empty(max(G, 1), device=dev, dtype=torch.int64)
    obj_idx = torch.empty(G, device=dev, dtype=torch.int64)
    tgt = torch.empty((G, 4), device=dev, dtype=torch.float32)
    noobj = torch.empty((bs, N), device=dev, dtype=torch.uint8)
    check(lib().mi355det_yolo_assign(C.byref(geom), ptr(boxes), ptr(off), bs, G, max(counts) if counts else 0, ptr(key),
                                     ptr(obj_idx), ptr(tgt), ptr(noobj), stream_ptr()), "yolo_assign")
    return obj_idx, tgt, noobj


def yolo_loss(geom, cfg, hviews, gviews, off, labels, obj_idx, tgt, noobj, idf, bs, G):
    dev = labels.device
    N = geom.off[geom.num_scales]
    wsb = lib().mi355det_yolo_loss_workspace(bs, N)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    out12 = torch.empty(12, device=dev, dtype=torch.float32)
    check(lib().mi355det_yolo_loss(C.byref(geom), C.byref(cfg), hviews, gviews, ptr(off), ptr(labels), ptr(obj_idx), ptr(tgt),
                                   ptr(noobj), ptr(idf), bs, G, ptr(ws), wsb, ptr(out12), stream_ptr()), "yolo_loss")
    return out12


def yolo_decode(geom, hviews, idf, bs, softmax_cls=True, want_scores=False):
    """-> decoded [bs,N,attrs] (and, with want_scores on channels-last heads, score [bs,N] + arg-max label [bs,N])."""
    N, attrs = geom.off[geom.num_scales], geom.num_classes + 5
    dev = torch.device("cuda", torch.cuda.current_device())
    out = torch.empty((bs, N, attrs), device=dev, dtype=torch.float32)
    score = label = None
    if want_scores and all(hviews[k].sc == 1 for k in range(geom.num_scales)):
        score = torch.empty((bs, N), device=dev, dtype=torch.float32)
        label = torch.empty((bs, N), device=dev, dtype=torch.int32)
    check(lib().mi355det_yolo_decode(C.byref(geom), hviews, ptr(idf), bs, int(softmax_cls), ptr(out), ptr(score), ptr(label), stream_ptr()),
          "yolo_decode")
    return (out, score, label) if want_scores else out


def yolo_candidates(pred, conf_thr, max_cand=None, score=None, label=None):
    """test_one_epoch.py:24-35 -> (cand [bs,max_cand,6], count [bs] i32).  score/label: the fused outputs of yolo_decode."""
    pred = _f32c(pred)
    bs, n, attrs = pred.shape
    max_cand = int(max_cand or min(n, 16384))
    cand = torch.empty((bs, max_cand, 6), device=pred.device, dtype=torch.float32)
    count = torch.empty(bs, device=pred.device, dtype=torch.int32)
    wsb = lib().mi355det_yolo_candidates_workspace(bs, n)
    ws = torch.empty(wsb, device=pred.device, dtype=torch.uint8)
    check(lib().mi355det_yolo_candidates(ptr(pred), ptr(score), ptr(label), bs, n, attrs, float(conf_thr), ptr(cand), ptr(count), max_cand,
                                         ptr(ws), wsb, stream_ptr()), "yolo_candidates")
    return cand, count


def nms_majority_batched(boxes, count, thresh_iou, num_classes):
    """boxes [bs,max_n,6], count [bs] i32 -> (rows [bs,max_n,6], idx [bs,max_n] i32, kept [bs] i32)."""
    boxes = _f32c(boxes)
    bs, max_n, _ = boxes.shape
    rows = torch.empty_like(boxes)
    idx = torch.empty((bs, max_n), device=boxes.device, dtype=torch.int32)
    kept = torch.empty(bs, device=boxes.device, dtype=torch.int32)
    wsb = lib().mi355det_nms_workspace(bs, max_n)
    ws = torch.empty(wsb, device=boxes.device, dtype=torch.uint8)
    check(lib().mi355det_nms_majority(ptr(boxes), ptr(count), bs, max_n, float(thresh_iou), int(num_classes), ptr(rows), ptr(idx),
                                      ptr(kept), ptr(ws), wsb, stream_ptr()), "nms_majority")
    return rows, idx, kept


# ------------------------------------------------------------------------------- torchvision side
def box_iou(boxes1, boxes2):
    b1, b2 = _f32c(boxes1), _f32c(boxes2)
    out = torch.empty((b1.shape[0], b2.shape[0]), device=b1.device, dtype=torch.float32)
    check(lib().mi355det_box_iou(ptr(b1), ptr(b2), ptr(out), b1.shape[0], b2.shape[0], stream_ptr()), "box_iou")
    return out


def nms_raw(boxes, scores, iou_threshold, idxs=None):
    """nms / batched_nms without the host round trip: (keep [n] int64 - only the first `count` entries are defined -, count [1] int32 on
    the device).  Callers that post-process several images read all their counts with ONE transfer."""
    boxes, scores = _f32c(boxes), _f32c(scores)
    n = boxes.shape[0]
    if idxs is not None:
        idxs = idxs.to(torch.int64).contiguous()
    keep = torch.zeros(max(n, 1), device=boxes.device, dtype=torch.int64)
    cnt = torch.zeros(1, device=boxes.device, dtype=torch.int32)
    if n:
        wsb = lib().mi355det_nms_workspace(1, n)
        ws = torch.empty(wsb, device=boxes.device, dtype=torch.uint8)
        check(lib().mi355det_nms(ptr(boxes), ptr(scores), ptr(idxs), n, float(iou_threshold), ptr(keep), ptr(cnt), ptr(ws), wsb,
                                 stream_ptr()), "nms")
    return keep, cnt


def nms_batch(boxes, scores, iou_threshold, idxs=None):
    """`bs` independent NMS problems of the same size in one launch sequence: boxes [bs,n,4], scores [bs,n], idxs [bs,n] (categories, as
    batched_nms) or None -> (keep [bs,n] int64 - the first counts[b] entries of row b are defined -, counts [bs] int32 on the device)."""
    boxes, scores = _f32c(boxes), _f32c(scores)
    bs, n = scores.shape
    if idxs is not None:
        idxs = idxs.to(torch.int64).contiguous()
    keep = torch.zeros((bs, max(n, 1)), device=boxes.device, dtype=torch.int64)
    cnt = torch.zeros(bs, device=boxes.device, dtype=torch.int32)
    if n and bs:
        wsb = lib().mi355det_nms_workspace(bs, n)
        ws = torch.empty(wsb, device=boxes.device, dtype=torch.uint8)
        check(lib().mi355det_nms_batch(ptr(boxes), ptr(scores), ptr(idxs), bs, n, float(iou_threshold), ptr(keep), ptr(cnt), ptr(ws), wsb,
                                       stream_ptr()), "nms_batch")
    return keep, cnt


def rpn_proposals(objectness, deltas, anchors, clip_limits, level_counts, pre_nms_top_n, post_nms_top_n, nms_thresh, score_thresh=0.0,
                  min_size=1e-3, xform_clip=math.log(1000.0 / 16), counts_out=None):
    """RegionProposalNetwork.filter_proposals (rpn.py:215-280) incl. the decode of the selected anchors, whole batch, one host call:
    objectness [N,A] logits, deltas [N,A,4], anchors [A,4], clip_limits [N,4] = (w,h,w,h) -> (boxes [N,post,4], scores [N,post],
    counts [N] int32 on the device; rows beyond counts[i] are zero)."""
    objectness, deltas, anchors, clip_limits = _f32c(objectness), _f32c(deltas), _f32c(anchors), _f32c(clip_limits)
    n, a = objectness.shape
    if deltas.numel() != n * a * 4 or anchors.numel() != a * 4 or clip_limits.numel() != n * 4 or sum(level_counts) != a:
        raise ValueError("rpn_proposals: objectness [N,A], deltas [N,A,4], anchors [A,4], clip_limits [N,4], sum(level_counts) == A")
    lc = (C.c_int64 * len(level_counts))(*[int(v) for v in level_counts])
    wsb = lib().mi355det_rpn_proposals_workspace(n, lc, len(level_counts), int(pre_nms_top_n))
    if wsb == 0:
        raise ValueError("rpn_proposals: need 1..8 non-empty levels, a positive batch and pre_nms_top_n")
    dev = objectness.device
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    boxes = torch.empty((n, int(post_nms_top_n), 4), device=dev, dtype=torch.float32)
    scores = torch.empty((n, int(post_nms_top_n)), device=dev, dtype=torch.float32)
    counts = counts_out if counts_out is not None else torch.empty(n, device=dev, dtype=torch.int32)
    check(lib().mi355det_rpn_proposals(ptr(objectness), ptr(deltas), ptr(anchors), ptr(clip_limits), n, lc, len(level_counts), int(pre_nms_top_n),
                                       int(post_nms_top_n), float(nms_thresh), float(score_thresh), float(min_size), float(xform_clip),
                                       ptr(boxes), ptr(scores), ptr(counts), ptr(ws), wsb, stream_ptr()), "rpn_proposals")
    return boxes, scores, counts


def retina_detections(cls_logits_per_level, bbox_reg_per_level, anchors_per_level, clip_limits, logit_thresh, topk_candidates, nms_thresh,
                      detections_per_img, xform_clip=math.log(1000.0 / 16)):
    """RetinaNet.postprocess_detections (retinanet.py:414-472), whole batch, one host call: per level cls_logits [N,HWA,K] (fp32, tf-idf
    scaling already applied), bbox regression [N,HWA,4], anchors [HWA,4]; clip_limits [N,4] = (w,h,w,h) ->
    (boxes [N,det,4], scores [N,det], labels [N,det] i64, counts [N] i32 on the device)."""
    cl = [_f32c(t) for t in cls_logits_per_level]
    rg = [_f32c(t) for t in bbox_reg_per_level]
    an = [_f32c(t) for t in anchors_per_level]
    clip_limits = _f32c(clip_limits)
    nl, n, k = len(cl), cl[0].shape[0], cl[0].shape[-1]
    hwa = [int(t.shape[1]) for t in cl]
    if not 1 <= nl <= 8 or any(r.shape[:2] != c.shape[:2] or a.shape[0] != c.shape[1] for c, r, a in zip(cl, rg, an)) or clip_limits.numel() != 4 * n:
        raise ValueError("retina_detections: 1..8 levels of cls_logits [N,HWA,K], bbox_regression [N,HWA,4], anchors [HWA,4]; clip_limits [N,4]")
    i64a, vpa = C.c_int64 * nl, C.c_void_p * nl
    la = i64a(*hwa)
    wsb = lib().mi355det_retina_detections_workspace(n, la, nl, k, int(topk_candidates))
    if wsb == 0:
        raise ValueError("retina_detections: need fewer than 2^32 scores per image and level and 1 <= topk_candidates <= 16384")
    dev = cl[0].device
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    det = int(detections_per_img)
    boxes = torch.empty((n, det, 4), device=dev, dtype=torch.float32)
    scores = torch.empty((n, det), device=dev, dtype=torch.float32)
    labels = torch.empty((n, det), device=dev, dtype=torch.int64)
    counts = torch.empty(n, device=dev, dtype=torch.int32)
    check(lib().mi355det_retina_detections(vpa(*[ptr(t) for t in cl]), vpa(*[ptr(t) for t in rg]), vpa(*[ptr(t) for t in an]), la, nl, n, k,
                                           ptr(clip_limits), float(logit_thresh), int(topk_candidates), float(nms_thresh), det, float(xform_clip),
                                           ptr(boxes), ptr(scores), ptr(labels), ptr(counts), ptr(ws), wsb, stream_ptr()), "retina_detections")
    return boxes, scores, labels, counts


def roi_detections(scores, box_regression, proposals, clip_limits, score_thresh, max_candidates, weights, nms_thresh, detections_per_img,
                   min_size=1e-2, xform_clip=math.log(1000.0 / 16), meta_out=None):
    """RoIHeads.postprocess_detections (roi_heads.py:715-781), whole batch, one host call: scores [N,P,C] (class 0 / padded proposals already
    below the threshold), box_regression [N,P,C*4], proposals [N,P,4], clip_limits [N,4] -> (boxes [N,det,4], scores [N,det], labels [N,det],
    meta [2N] i32 = detections per image, then candidates per image (== max_candidates: possibly truncated))."""
    scores, box_regression, proposals, clip_limits = _f32c(scores), _f32c(box_regression), _f32c(proposals), _f32c(clip_limits)
    n, p, c = scores.shape
    k = int(min(max_candidates, p * c))
    if box_regression.numel() != n * p * c * 4 or proposals.numel() != n * p * 4 or clip_limits.numel() != n * 4:
        raise ValueError("roi_detections: scores [N,P,C], box_regression [N,P,C*4], proposals [N,P,4], clip_limits [N,4]")
    wsb = lib().mi355det_roi_detections_workspace(n, p, c, k)
    if wsb == 0:
        raise ValueError("roi_detections: need 1 <= max_candidates <= 16384 and fewer than 2^32 scores per image")
    dev = scores.device
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    det = int(detections_per_img)
    boxes = torch.empty((n, det, 4), device=dev, dtype=torch.float32)
    out_s = torch.empty((n, det), device=dev, dtype=torch.float32)
    labels = torch.empty((n, det), device=dev, dtype=torch.int64)
    meta = meta_out if meta_out is not None else torch.empty(2 * n, device=dev, dtype=torch.int32)
    check(lib().mi355det_roi_detections(ptr(scores), ptr(box_regression), ptr(proposals), ptr(clip_limits), n, p, c, float(score_thresh), k,
                                        *[float(w) for w in weights], float(xform_clip), float(min_size), float(nms_thresh), det, ptr(boxes),
                                        ptr(out_s), ptr(labels), ptr(meta[:n]), ptr(meta[n:]), ptr(ws), wsb, stream_ptr()), "roi_detections")
    return boxes, out_s, labels, meta, k


def rpn_loss(objectness, pred_bbox_deltas, labels, regression_targets, pos_idx, sampled_idx):
    """RegionProposalNetwork.compute_loss (rpn.py:282-318) with its gradients, one launch: objectness [T(,1)], deltas / targets [T,4],
    labels [T] float, pos_idx / sampled_idx int64 -> (losses [2] = (objectness, box), grad_objectness like objectness, grad_deltas [T,4])."""
    obj, dl, lab, tg = _f32c(objectness), _f32c(pred_bbox_deltas), _f32c(labels), _f32c(regression_targets)
    t = obj.numel()
    if dl.numel() != 4 * t or lab.numel() != t or tg.numel() != 4 * t or sampled_idx.numel() == 0:
        raise ValueError("rpn_loss: objectness [T], deltas / targets [T,4], labels [T], at least one sampled anchor")
    pos_idx, sampled_idx = pos_idx.to(torch.int64).contiguous(), sampled_idx.to(torch.int64).contiguous()
    losses = torch.empty(2, device=obj.device, dtype=torch.float32)
    g_obj, g_dl = torch.empty_like(obj), torch.empty_like(dl)
    check(lib().mi355det_rpn_loss(ptr(obj), ptr(dl), ptr(lab), ptr(tg), t, ptr(pos_idx) if pos_idx.numel() else None, pos_idx.numel(),
                                  ptr(sampled_idx), sampled_idx.numel(), ptr(losses), ptr(g_obj), ptr(g_dl), stream_ptr()), "rpn_loss")
    return losses, g_obj, g_dl


def roi_match(proposals, proposal_counts, gt_boxes, gt_labels, gt_offsets, fg_iou_thresh, bg_iou_thresh, counts_out=None):
    """RoIHeads.assign_targets_to_proposals over add_gt_proposals (roi_heads.py:627-652,664-668) for the whole batch, one launch: proposals
    [N,P,4] padded + proposal_counts [N] i32 on the device, gt_boxes [G,4] / gt_labels [G] concatenated, gt_offsets host list [N+1] ->
    (matched [N,C] i32, labels [N,C] i32, counts [N,2] i32 = positives / negatives), C = P + the largest ground-truth count."""
    proposals, gt_boxes = _f32c(proposals), _f32c(gt_boxes)
    n, p = proposals.shape[0], proposals.shape[1]
    if len(gt_offsets) != n + 1 or any(b <= a for a, b in zip(gt_offsets, gt_offsets[1:])):
        raise ValueError("No ground-truth boxes available for one of the images during training")          # Matcher, _utils.py:282-291
    c = p + max(b - a for a, b in zip(gt_offsets, gt_offsets[1:]))
    dev = proposals.device
    matched = torch.empty((n, c), device=dev, dtype=torch.int32)
    labels = torch.empty((n, c), device=dev, dtype=torch.int32)
    counts = counts_out if counts_out is not None else torch.empty((n, 2), device=dev, dtype=torch.int32)
    offs = (C.c_int32 * (n + 1))(*[int(v) for v in gt_offsets])
    check(lib().mi355det_roi_match(ptr(proposals), ptr(proposal_counts), n, p, ptr(gt_boxes), ptr(gt_labels.to(torch.int64).contiguous()), offs,
                                   float(fg_iou_thresh), float(bg_iou_thresh), c, ptr(matched), ptr(labels), ptr(counts), stream_ptr()), "roi_match")
    return matched, labels, counts


def roi_sample(proposals, proposal_counts, gt_boxes, gt_offsets, matched, labels, perm_pos, perm_neg, num_pos, num_neg, weights):
    """BalancedPositiveNegativeSampler's selection from its `randperm` draws + the gathers and BoxCoder.encode of select_training_samples
    (roi_heads.py:654-713), whole batch, one launch -> (rois [S,5], labels [S] i64, matched [S] i64, regression_targets [S,4])."""
    n, p = proposals.shape[0], proposals.shape[1]
    total = int(sum(num_pos) + sum(num_neg))
    dev = proposals.device
    rois = torch.empty((total, 5), device=dev, dtype=torch.float32)
    out_l = torch.empty(total, device=dev, dtype=torch.int64)
    out_m = torch.empty(total, device=dev, dtype=torch.int64)
    reg = torch.empty((total, 4), device=dev, dtype=torch.float32)
    vpa, i32a = C.c_void_p * n, C.c_int32 * n
    offs = (C.c_int32 * (n + 1))(*[int(v) for v in gt_offsets])
    check(lib().mi355det_roi_sample(ptr(proposals), ptr(proposal_counts), n, p, ptr(gt_boxes), offs, matched.shape[1], ptr(matched), ptr(labels),
                                    vpa(*[ptr(t) for t in perm_pos]), vpa(*[ptr(t) for t in perm_neg]), i32a(*[int(v) for v in num_pos]),
                                    i32a(*[int(v) for v in num_neg]), *[float(w) for w in weights], ptr(rois), ptr(out_l), ptr(out_m), ptr(reg),
                                    stream_ptr()), "roi_sample")
    return rois, out_l, out_m, reg


def nms(boxes, scores, iou_threshold, idxs=None):
    boxes, scores = _f32c(boxes), _f32c(scores)
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, device=boxes.device, dtype=torch.int64)
    if idxs is not None:
        idxs = idxs.to(torch.int64).contiguous()
    keep = torch.empty(n, device=boxes.device, dtype=torch.int64)
    cnt = torch.empty(1, device=boxes.device, dtype=torch.int32)
    wsb = lib().mi355det_nms_workspace(1, n)
    ws = torch.empty(wsb, device=boxes.device, dtype=torch.uint8)
    check(lib().mi355det_nms(ptr(boxes), ptr(scores), ptr(idxs), n, float(iou_threshold), ptr(keep), ptr(cnt), ptr(ws), wsb,
                             stream_ptr()), "nms")
    return keep[: int(cnt.item())]


def match_anchors(gt, anchors, high, low, allow_low_quality):
    gt, anchors = _f32c(gt), _f32c(anchors)
    m, n = gt.shape[0], anchors.shape[0]
    if m == 0:
        raise ValueError("No ground-truth boxes available for one of the images during training")
    if n == 0:
        raise ValueError("No proposal boxes available for one of the images during training")
    best = torch.empty(m, device=gt.device, dtype=torch.int32)
    out = torch.empty(n, device=gt.device, dtype=torch.int64)
    check(lib().mi355det_match_anchors(ptr(gt), ptr(anchors), m, n, float(high), float(low), int(allow_low_quality), ptr(best),
                                       ptr(out), stream_ptr()), "match_anchors")
    return out


def box_encode(reference_boxes, proposals, weights):
    r, p = _f32c(reference_boxes), _f32c(proposals)
    out = torch.empty_like(p)
    check(lib().mi355det_box_encode(ptr(r), ptr(p), ptr(out), p.shape[0], *[float(w) for w in weights], stream_ptr()), "box_encode")
    return out


def box_decode(rel_codes, boxes, weights, clip):
    c, b = _f32c(rel_codes), _f32c(boxes)
    n, k = b.shape[0], c.shape[1] // 4
    out = torch.empty_like(c)
    check(lib().mi355det_box_decode(ptr(c), ptr(b), ptr(out), n, k, *[float(w) for w in weights], float(clip), stream_ptr()),
          "box_decode")
    return out


def anchor_grid(cell, gh, gw, sh, sw):
    cell = _f32c(cell)
    out = torch.empty((gh * gw * cell.shape[0], 4), device=cell.device, dtype=torch.float32)
    check(lib().mi355det_anchor_grid(ptr(cell), cell.shape[0], gh, gw, int(sh), int(sw), ptr(out), stream_ptr()), "anchor_grid")
    return out


def sigmoid_focal_loss_sum(x, t, alpha, gamma, scale=None, valid=None, want_grad=True, grad_scale=1.0):
    """sum-reduced loss and its gradient wrt x in one pass: x,t [rows,k]."""
    x, t = _f32c(x), _f32c(t)
    rows, k = (x.shape[0], x.shape[1]) if x.dim() == 2 else (x.numel(), 1)
    loss = torch.zeros(1, device=x.device, dtype=torch.float32)
    grad = torch.empty_like(x) if want_grad else None
    if valid is not None:
        valid = valid.to(torch.uint8).contiguous()
    check(lib().mi355det_sigmoid_focal_loss(ptr(x), ptr(t), ptr(scale), ptr(valid), rows, k, float(alpha), float(gamma),
                                            float(grad_scale), ptr(loss), ptr(grad), stream_ptr()), "sigmoid_focal_loss")
    return loss[0], grad


def sigmoid_focal_loss_elem(x, t, alpha, gamma, want_grad=True):
    """Unreduced focal loss (reduction='none') and its derivative, elementwise on any shape."""
    x, t = _f32c(x), _f32c(t)
    if x.shape != t.shape:
        raise ValueError("inputs and targets must have the same shape")
    loss = torch.empty_like(x)
    grad = torch.empty_like(x) if want_grad else None
    check(lib().mi355det_sigmoid_focal_loss_elem(ptr(x), ptr(t), x.numel(), float(alpha), float(gamma), ptr(loss), ptr(grad), stream_ptr()),
          "sigmoid_focal_loss_elem")
    return loss, grad


def retina_cls_loss_sum(logits, matched, gt_labels, alpha, gamma, scale=None, want_grad=True, grad_scale=1.0):
    logits = _f32c(logits)
    rows, k = logits.shape
    loss = torch.zeros(1, device=logits.device, dtype=torch.float32)
    grad = torch.empty_like(logits) if want_grad else None
    check(lib().mi355det_retina_cls_loss(ptr(logits), ptr(matched.contiguous()), ptr(gt_labels.to(torch.int64).contiguous()),
                                         ptr(scale), rows, k, float(alpha), float(gamma), float(grad_scale), ptr(loss), ptr(grad),
                                         stream_ptr()), "retina_cls_loss")
    return loss[0], grad


# ------------------------------------------------------------------------------------ conv path
def conv_shape(n, h, w, cin, cout, ksize, stride, in_ld=None, out_ld=None):
    pad = (ksize - 1) // 2
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    return _lib.ConvShape(n, h, w, cin, ho, wo, cout, ksize, stride, pad, in_ld or cin, out_ld or cout)


def pad_to(v, m):
    return (v + m - 1) // m * m


def cout_pad_of(cout):
    if cout >= 2048:
        return pad_to(cout, 256)        # very wide outputs (the 1204-class RetinaNet head: 10 836): whole 256-channel tiles for igemm8
    return pad_to(cout, 128) if cout >= 128 or cout % 32 else cout


def pack_weights(shape, w_master, want_dgrad=True, ohwi=False, wf=None, wd=None):
    """fp32 [cout,cin,k,k] (or OHWI [cout,k,k,cin]) -> (bf16 fwd pack [cout_pad, k*k*cin], bf16 dgrad pack)."""
    dev = w_master.device
    cp = cout_pad_of(shape.cout)
    kk = shape.ksize * shape.ksize
    if wf is None:
        wf = torch.empty(cp * kk * shape.cin, device=dev, dtype=act_dtype())
    if want_dgrad and wd is None:
        wd = torch.empty(lib().mi355det_dgrad_pack_elems(C.byref(shape)), device=dev, dtype=act_dtype())
    check(lib().mi355det_pack_weights(C.byref(shape), ptr(w_master.contiguous()), int(ohwi), ptr(wf), cp, ptr(wd), stream_ptr()),
          "pack_weights")
    return wf, wd


def conv_fwd(shape, x, w_fwd, y, bias=None, out_f32=False, stats=None):
    cp = cout_pad_of(shape.cout)
    check(lib().mi355det_conv_fwd(C.byref(shape), ptr(x), ptr(w_fwd), ptr(bias), ptr(y), int(out_f32), ptr(stats), cp, stream_ptr()),
          "conv_fwd")


def conv_stats_rows(shape):
    return lib().mi355det_conv_stats_rows(C.byref(shape), cout_pad_of(shape.cout))


def conv_dgrad(shape, dy, w_dgrad, dx, residual=None, residual_ld=0):
    check(lib().mi355det_conv_dgrad(C.byref(shape), ptr(dy), ptr(w_dgrad), ptr(dx), ptr(residual), int(residual_ld), stream_ptr()),
          "conv_dgrad")


_WGRAD_WS = {}


def conv_wgrad(shape, x, dy, dw, dbias=None, workspace=None):
    need = lib().mi355det_conv_wgrad_workspace(C.byref(shape))
    if workspace is None and need:
        key = dw.device
        if key not in _WGRAD_WS or _WGRAD_WS[key].numel() < need:
            _WGRAD_WS[key] = torch.empty(need, device=dw.device, dtype=torch.uint8)
        workspace = _WGRAD_WS[key]
    check(lib().mi355det_conv_wgrad(C.byref(shape), ptr(x), ptr(dy), ptr(dw), ptr(dbias), ptr(workspace),
                                    workspace.numel() if workspace is not None else 0, stream_ptr()), "conv_wgrad")


# ------------------------------------------------------------------------------------ RoIAlign / top-k
def roi_align_multi(feats, rois, output_size, scales, sampling_ratio=2, aligned=False, k_min=2, k_max=5, grad_out=None):
    """feats: list of NCHW fp32 maps (1..4 levels); rois [K,5].  Forward -> [K,C,ph,pw]; with grad_out -> list of dfeats."""
    feats = [_f32c(f) for f in feats]
    rois = _f32c(rois)
    nl = len(feats)
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    K, Cc = rois.shape[0], feats[0].shape[1]
    P = (C.c_void_p * nl)(*[f.data_ptr() for f in feats])
    hs = (C.c_int32 * nl)(*[f.shape[2] for f in feats])
    ws = (C.c_int32 * nl)(*[f.shape[3] for f in feats])
    sc = (C.c_float * nl)(*[float(s) for s in scales])
    if grad_out is None:
        out = torch.empty((K, Cc, ph, pw), device=rois.device, dtype=torch.float32)
        check(lib().mi355det_roi_align(P, hs, ws, sc, nl, ptr(rois), K, Cc, ph, pw, int(sampling_ratio), int(aligned), k_min, k_max,
                                       ptr(out), None, None, stream_ptr()), "roi_align")
        return out
    g = _f32c(grad_out)
    dfs = [torch.zeros_like(f) for f in feats]
    G = (C.c_void_p * nl)(*[d.data_ptr() for d in dfs])
    check(lib().mi355det_roi_align(P, hs, ws, sc, nl, ptr(rois), K, Cc, ph, pw, int(sampling_ratio), int(aligned), k_min, k_max,
                                   None, ptr(g), G, stream_ptr()), "roi_align_bwd")
    return dfs


def topk_rows(x, k, min_value=float("-inf")):
    """x [rows,n] fp32 -> (values [rows,k], indices [rows,k] i64, count [rows] i32), descending, ties lower index first."""
    x = _f32c(x)
    rows, n = x.shape
    k = int(min(k, n))
    idx = torch.zeros((rows, k), device=x.device, dtype=torch.int64)
    val = torch.zeros((rows, k), device=x.device, dtype=torch.float32)
    cnt = torch.empty(rows, device=x.device, dtype=torch.int32)
    if n >= 65536:          # long rows (flattened HWA x K score maps): several workgroups per row
        wsb = lib().mi355det_topk_workspace(rows)
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
        check(lib().mi355det_topk_ws(ptr(x), rows, n, x.stride(0), k, float(min_value), ptr(idx), ptr(val), ptr(cnt), ptr(ws), wsb, stream_ptr()), "topk")
        return val, idx, cnt
    check(lib().mi355det_topk(ptr(x), rows, n, x.stride(0), k, float(min_value), ptr(idx), ptr(val), ptr(cnt), stream_ptr()), "topk")
    return val, idx, cnt


def topk_segments(x, seg_counts, k, min_value=float("-inf")):
    """x [rows, sum(seg_counts)] fp32: top-min(k, n_s) of every segment of every row in one call (RegionProposalNetwork._get_top_n_idx,
    rpn.py:215-228) -> list per segment of (values [rows,k_s], indices within the segment [rows,k_s] i64, count [rows] i32)."""
    x = _f32c(x)
    rows, a = x.shape
    if sum(seg_counts) != a or not 1 <= len(seg_counts) <= 8:
        raise ValueError("topk_segments: 1..8 segments that add up to the row length")
    ns = len(seg_counts)
    ks = [int(min(k, n)) for n in seg_counts]
    starts = [sum(seg_counts[:i]) for i in range(ns)]
    outs = [(torch.zeros((rows, kk), device=x.device, dtype=torch.float32), torch.zeros((rows, kk), device=x.device, dtype=torch.int64),
             torch.empty(rows, device=x.device, dtype=torch.int32)) for kk in ks]
    wsb = lib().mi355det_topk_workspace(rows)
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
    i64a, i32a, vpa = C.c_int64 * ns, C.c_int32 * ns, C.c_void_p * ns
    check(lib().mi355det_topk_segments(ptr(x), rows, x.stride(0), ns, i64a(*starts), i64a(*[int(n) for n in seg_counts]), i32a(*ks), float(min_value),
                                       vpa(*[ptr(o[1]) for o in outs]), vpa(*[ptr(o[0]) for o in outs]), vpa(*[ptr(o[2]) for o in outs]),
                                       ptr(ws), wsb, stream_ptr()), "topk_segments")
    return outs


# ------------------------------------------------------------------------------------ ResNet-FPN / RetinaNet companions
def conv_fwd_ex(shape, x, w_fwd, y, scale=None, shift=None, residual=None, residual_ld=0, relu=False, out_f32=False, out_image_stride=0,
                leaky_slope=None):
    """y = relu?(conv(x,w)*scale + shift + residual): FrozenBatchNorm2d / bias / identity add fused into the conv epilogue.
    leaky_slope: Darknet form y = lrelu(conv*scale + shift) + residual."""
    mode = 2 if leaky_slope is not None else int(bool(relu))
    e = _lib.ConvEpilogue(ptr(scale), ptr(shift), ptr(residual), int(residual_ld), mode, int(out_image_stride), float(leaky_slope or 0.0))
    check(lib().mi355det_conv_fwd_ex(C.byref(shape), ptr(x), ptr(w_fwd), C.byref(e), ptr(y), int(out_f32), cout_pad_of(shape.cout), stream_ptr()),
          "conv_fwd_ex")


def im2col_nchw(img, ksize, stride, pad, kpad, mean=None, inv_std=None):
    n, c, h, w = img.shape
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    out = torch.empty((n, ho, wo, kpad), device=img.device, dtype=torch.bfloat16)
    check(lib().mi355det_im2col_nchw(ptr(_f32c(img)), ptr(mean), ptr(inv_std), ptr(out), n, c, h, w, ksize, stride, pad, kpad, stream_ptr()),
          "im2col_nchw")
    return out


def maxpool3x3s2(x):
    n, h, w, c = x.shape
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), device=x.device, dtype=torch.bfloat16)
    check(lib().mi355det_maxpool3x3s2(ptr(x), c, n, h, w, c, ptr(out), c, stream_ptr()), "maxpool3x3s2")
    return out


def relu_affine_bwd(g1, a, scale=None, g2=None, relu=True, want_gm=False):
    c = g1.shape[-1]
    pixels = g1.numel() // c
    dz = torch.empty_like(g1)
    gm = torch.empty_like(g1) if want_gm else None
    check(lib().mi355det_relu_affine_bwd(ptr(g1), c, ptr(g2), c if g2 is not None else 0, ptr(a), c, ptr(scale), c, pixels, int(relu), ptr(dz), c,
                                         ptr(gm), c, stream_ptr()), "relu_affine_bwd")
    return (dz, gm) if want_gm else dz


def upsample_nearest_add(x, lateral, out_hw):
    n, h, w, c = x.shape
    out = torch.empty((n, out_hw[0], out_hw[1], c), device=x.device, dtype=torch.bfloat16)
    check(lib().mi355det_upsample_nearest_add(ptr(x), c, n, h, w, c, ptr(lateral), c if lateral is not None else 0, out_hw[0], out_hw[1], ptr(out), c,
                                              stream_ptr()), "upsample_nearest_add")
    return out


def upsample_nearest_bwd(g, hw, accumulate=None):
    n, gh, gw, c = g.shape
    out = torch.empty((n, hw[0], hw[1], c), device=g.device, dtype=torch.bfloat16)
    check(lib().mi355det_upsample_nearest_bwd(ptr(g), c, n, hw[0], hw[1], c, gh, gw, ptr(accumulate), c if accumulate is not None else 0, ptr(out), c,
                                              stream_ptr()), "upsample_nearest_bwd")
    return out


def retina_loss(cls_logits, bbox_regression, anchors, matched, gt_boxes, gt_labels, gt_offsets, k=None, class_scale=None, alpha=0.25, gamma=2.0,
                want_grad=True, grad_scale=1.0, grad_logits=None, grad_regression=None, cls_levels=None, anchors_per_pixel=None):
    """Batched RetinaNetHead.compute_loss.  -> (losses[2] = (classification, bbox_regression), num_fg [N], grad_logits, grad_regression).
    cls_levels (list of bf16 [n, h, w, ld] buffers, one per pyramid level, + anchors_per_pixel): the class gradient is written there in the
    head convolution's layout instead of an fp32 grad_logits (returned as None)."""
    n, rows, k = cls_logits.shape
    dev = cls_logits.device
    losses = torch.empty(2, device=dev)
    nfg = torch.empty(n, device=dev)
    if cls_levels is not None:
        from ._lib import LevelGrads
        if len(cls_levels) > 8 or not anchors_per_pixel:
            raise ValueError("retina_loss: at most 8 levels, and anchors_per_pixel is required with cls_levels")
        lv = LevelGrads()
        lv.n_levels, lv.anchors_per_pixel = len(cls_levels), int(anchors_per_pixel)
        for q, t in enumerate(cls_levels):
            if t.dtype != torch.bfloat16 or t.dim() != 4 or t.shape[0] != n or not t.is_contiguous():
                raise ValueError("retina_loss: cls_levels must be contiguous bf16 [n, h, w, ld] tensors")
            lv.pixels[q], lv.grad[q], lv.grad_ld[q] = t.shape[1] * t.shape[2], t.data_ptr(), t.shape[3]
        grad_regression = torch.empty_like(bbox_regression) if grad_regression is None else grad_regression
        check(lib().mi355det_retina_loss_lv(ptr(cls_logits), ptr(bbox_regression), ptr(anchors), ptr(matched), ptr(gt_boxes), ptr(gt_labels),
                                            ptr(gt_offsets), ptr(class_scale), n, rows, k, float(alpha), float(gamma), float(grad_scale), ptr(nfg),
                                            ptr(losses), C.byref(lv), ptr(grad_regression), stream_ptr()), "retina_loss_lv")
        return losses, nfg, None, grad_regression
    if want_grad:
        grad_logits = torch.empty_like(cls_logits) if grad_logits is None else grad_logits
        grad_regression = torch.empty_like(bbox_regression) if grad_regression is None else grad_regression
    check(lib().mi355det_retina_loss(ptr(cls_logits), ptr(bbox_regression), ptr(anchors), ptr(matched), ptr(gt_boxes), ptr(gt_labels), ptr(gt_offsets),
                                     ptr(class_scale), n, rows, k, float(alpha), float(gamma), float(grad_scale), ptr(nfg), ptr(losses),
                                     ptr(grad_logits) if want_grad else None, ptr(grad_regression) if want_grad else None, stream_ptr()), "retina_loss")
    return losses, nfg, grad_logits, grad_regression


def conv_dgrad_bn(shape, dy, w_dgrad, dx, z, scale_shift, slope, residual=None, residual_ld=0):
    """dgrad + fused BN-backward partial sums of the layer that produced dx's activation -> sums [2*cin]."""
    L = lib()
    rows = L.mi355det_conv_dgrad_bn_rows(C.byref(shape))
    cin_pad = pad_to(shape.cin, 32)
    partials = torch.empty((rows + 64, 2, cin_pad), device=dx.device, dtype=torch.float32)
    check(L.mi355det_conv_dgrad_bn(C.byref(shape), ptr(dy), ptr(w_dgrad), ptr(dx), ptr(residual), int(residual_ld), ptr(z), shape.in_ld,
                                   ptr(scale_shift), float(slope), ptr(partials), stream_ptr()), "conv_dgrad_bn")
    sums = torch.empty(2 * shape.cin, device=dx.device, dtype=torch.float32)
    check(L.mi355det_bn_bwd_sum_partials(ptr(partials), rows, shape.cin, cin_pad, ptr(sums), stream_ptr()), "bn_bwd_sum_partials")
    return sums


def roi_align_nhwc(feats, rois, output_size, scales, sampling_ratio=2, aligned=False, k_min=2, k_max=5, grad_out=None):
    """Channels-last RoIAlign: feats = list of bf16 NHWC maps [n,h,w,C] (1..4 levels, pixel pitch = stride(2)); rois [K,5].
    Forward -> fp32 [K,C,ph,pw]; with grad_out -> list of fp32 NHWC feature gradients."""
    for f in feats:
        if f.dtype != torch.bfloat16 or f.dim() != 4 or f.stride(3) != 1:
            raise ValueError("roi_align_nhwc expects bf16 [n,h,w,C] tensors with contiguous channels")
    rois = _f32c(rois)
    nl = len(feats)
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    K, Cc = rois.shape[0], feats[0].shape[3]
    P = (C.c_void_p * nl)(*[f.data_ptr() for f in feats])
    hs = (C.c_int32 * nl)(*[f.shape[1] for f in feats])
    ws = (C.c_int32 * nl)(*[f.shape[2] for f in feats])
    lds = (C.c_int32 * nl)(*[f.stride(2) for f in feats])
    sc = (C.c_float * nl)(*[float(s) for s in scales])
    if grad_out is None:
        out = torch.empty((K, Cc, ph, pw), device=rois.device, dtype=torch.float32)
        check(lib().mi355det_roi_align_nhwc(P, hs, ws, lds, sc, nl, ptr(rois), K, Cc, ph, pw, int(sampling_ratio), int(aligned), k_min, k_max,
                                            ptr(out), None, None, stream_ptr()), "roi_align_nhwc")
        return out
    g = _f32c(grad_out)
    dfs = [torch.zeros((f.shape[0], f.shape[1], f.shape[2], Cc), device=f.device, dtype=torch.float32) for f in feats]
    G = (C.c_void_p * nl)(*[d.data_ptr() for d in dfs])
    check(lib().mi355det_roi_align_nhwc(P, hs, ws, lds, sc, nl, ptr(rois), K, Cc, ph, pw, int(sampling_ratio), int(aligned), k_min, k_max,
                                        None, ptr(g), G, stream_ptr()), "roi_align_nhwc_bwd")
    return dfs


FRCNN_LOSS_TYPES = {"ce": 0, "bce": 1, "focal_loss": 2, "gombit": 3, "gombit_fl": 4}


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets, class_scale=None, class_weights=None, loss_type="ce", want_grad=True):
    """roi_heads.py:24-96 (+ the tf-idf scaling of its call site :826-827) -> (losses [2], grad_logits, grad_box)."""
    if loss_type not in FRCNN_LOSS_TYPES:
        raise ValueError(f"fastrcnn_loss: unknown loss_type {loss_type!r} (reference: 'ce', 'bce', 'focal_loss', 'gombit', 'gombit_fl')")
    x, b = _f32c(class_logits), _f32c(box_regression)
    n, k = x.shape
    if b.shape != (n, 4 * k):
        raise ValueError("fastrcnn_loss: box_regression must be [n, 4*k]")
    lab = labels.to(torch.int64).contiguous()
    tgt = _f32c(regression_targets)
    cs = None if class_scale is None else _f32c(class_scale.reshape(-1))
    cw = None if class_weights is None else _f32c(class_weights.reshape(-1))
    L = lib()
    wsb = L.mi355det_fastrcnn_loss_workspace(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    losses = torch.empty(2, dtype=torch.float32, device=x.device)
    gl = torch.empty_like(x) if want_grad else None
    gb = torch.empty_like(b) if want_grad else None
    check(L.mi355det_fastrcnn_loss(ptr(x), ptr(b), ptr(lab), ptr(tgt), ptr(cs), ptr(cw), n, k, FRCNN_LOSS_TYPES[loss_type], ptr(losses), ptr(gl),
                                   ptr(gb), ptr(ws), wsb, stream_ptr()), "fastrcnn_loss")
    return losses, gl, gb
