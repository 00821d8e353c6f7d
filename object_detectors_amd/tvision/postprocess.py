"""Mirrors of the proposal / detection post-processing of the torchvision_models path, composed from the HIP kernels:
  RegionProposalNetwork._get_top_n_idx + filter_proposals   tvision/rpn.py:215-280
  RetinaNet.postprocess_detections                          tvision/retinanet.py:414-472
  RoIHeads.postprocess_detections                           tvision/roi_heads.py:715-781
Only index bookkeeping (gather by the selected indices, concatenation, ragged python lists) stays in torch."""
import math

import torch

from .. import ops
from . import boxes as box_ops
from ._utils import BoxCoder


_CLIP_LIMITS = {}
_RETINA_FUSED = True      # False (tests monkeypatch it): the torch-composed chain, the bit-exact reference of mi355det_retina_detections


def _clip_limits(image_shapes, device, dtype):
    """[N, 1, 4] tensor (w, h, w, h) of the image sizes; cached per size tuple: building it from a Python list is a blocking host-to-device
    copy in the middle of the proposal filter (0.9 ms of a 14.8 ms step waiting for the device)."""
    key = (tuple((int(s[0]), int(s[1])) for s in image_shapes), str(device), dtype)
    t = _CLIP_LIMITS.get(key)
    if t is None:
        if len(_CLIP_LIMITS) > 64:
            _CLIP_LIMITS.clear()
        host = torch.tensor([[float(s[1]), float(s[0])] * 2 for s in image_shapes], dtype=dtype)
        if torch.device(device).type == "cuda":
            host = host.pin_memory()                       # a miss (new image sizes) costs an asynchronous copy, not a device drain
        t = _CLIP_LIMITS[key] = host.to(device, non_blocking=True)[:, None, :]
    return t


def rpn_filter_proposals(proposals, objectness, image_shapes, num_anchors_per_level, pre_nms_top_n, post_nms_top_n,
                         nms_thresh=0.7, score_thresh=0.0, min_size=1e-3):
    """proposals [N, A, 4] decoded boxes, objectness [N, A] logits -> (list of boxes [<=post,4], list of scores)."""
    num_images = proposals.shape[0]
    objectness = objectness.detach().reshape(num_images, -1)
    idx_parts, lvl_parts, off = [], [], 0
    for li, n in enumerate(num_anchors_per_level):
        k = min(pre_nms_top_n, n)
        _v, idx, _c = ops.topk_rows(objectness[:, off:off + n], k)          # per-level top-k for every image at once
        idx_parts.append(idx + off)
        lvl_parts.append(torch.full((k,), li, dtype=torch.int64, device=proposals.device))
        off += n
    top_idx = torch.cat(idx_parts, dim=1)
    levels = torch.cat(lvl_parts).unsqueeze(0).expand(num_images, -1)
    batch = torch.arange(num_images, device=proposals.device)[:, None]
    obj = torch.sigmoid(objectness[batch, top_idx])
    props = proposals[batch, top_idx]
    # rpn.py:259-280 per image: clip, drop boxes smaller than min_size and scores below score_thresh, per-level NMS, first post_nms_top_n.
    # The two filters run as a MASK here (score -> -inf) instead of a compaction: a dropped box ranks below every surviving one, so it can
    # never suppress one, and in the kept list (descending score) the survivors come first - the result is the reference's, but nothing has
    # a data-dependent shape until the very end, and the counts of ALL images are read with one device-to-host transfer (the per-image
    # compactions cost three synchronisations per image: 3.1 ms of a 17 ms step at batch 4).
    lim = _clip_limits(image_shapes, props.device, props.dtype)                                                        # [N, 1, 4] = w, h, w, h
    boxes = torch.minimum(props.clamp(min=0), lim)
    ws, hs = boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]
    valid = (ws >= min_size) & (hs >= min_size) & (obj >= score_thresh)
    masked = torch.where(valid, obj, torch.full_like(obj, float("-inf")))
    # all images in ONE launch sequence (sort / mask / scan per image side by side in the grid)
    keeps, cnt = ops.nms_batch(boxes, masked, nms_thresh, idxs=levels)
    ar = torch.arange(boxes.shape[1], device=props.device)[None, :]
    good = (ar < cnt[:, None]) & torch.gather(valid, 1, keeps.clamp(max=boxes.shape[1] - 1))
    counts = good.sum(1).clamp(max=post_nms_top_n).tolist()                   # the one synchronisation of the proposal filter
    final_boxes, final_scores = [], []
    for i in range(num_images):
        k = keeps[i][:counts[i]]
        final_boxes.append(boxes[i][k])
        final_scores.append(obj[i][k])
    return final_boxes, final_scores


def rpn_proposals_fused(deltas, objectness, anchors, image_shapes, num_anchors_per_level, pre_nms_top_n, post_nms_top_n, nms_thresh=0.7,
                        score_thresh=0.0, min_size=1e-3, xform_clip=math.log(1000.0 / 16), counts_out=None):
    """rpn.py:336-351 + filter_proposals (:215-280) from the RAW deltas [N, A, 4] and logits [N, A] in one host call
    (`mi355det_rpn_proposals`: only the selected anchors are decoded) and one device-to-host read; same results as
    `ops.box_decode` + `rpn_filter_proposals`, bit for bit (tests/test_gpu_proposals.py).  -> (list of boxes [<=post,4], list of scores)."""
    n = objectness.shape[0]
    lim = _clip_limits(image_shapes, objectness.device, torch.float32).reshape(n, 4)
    boxes, scores, counts = ops.rpn_proposals(objectness.detach().reshape(n, -1), deltas.detach().reshape(n, -1, 4), anchors, lim,
                                              list(num_anchors_per_level), pre_nms_top_n, post_nms_top_n, nms_thresh, score_thresh, min_size,
                                              xform_clip, counts_out=counts_out)
    if counts_out is not None:              # padded form: (boxes [N, post, 4], scores [N, post]); the caller reads counts_out when it needs to
        return boxes, scores
    counts = counts.tolist()                                                  # the one synchronisation of the proposal filter
    return [boxes[i, :c] for i, c in enumerate(counts)], [scores[i, :c] for i, c in enumerate(counts)]


def retinanet_postprocess_detections(cls_logits_per_level, bbox_reg_per_level, anchors_per_level, image_shapes, tfidf_post=None,
                                     score_thresh=0.05, topk_candidates=1000, nms_thresh=0.5, detections_per_img=300):
    """cls_logits_per_level: list of [N, HWA, K]; bbox_reg_per_level: list of [N, HWA, 4]; anchors_per_level: list of [HWA, 4].
    retinanet.py:414-472 for the WHOLE batch with one device-to-host read: per level the thresholded top-k of every image in one call, the
    candidates below the threshold kept as masked entries (score -inf, box 0: they rank last, never suppress anything, and do not move the
    per-category offsets of batched_nms), all images in one NMS launch sequence.  (The per-image / per-level form read a count back 5 times
    per image: 13.8 ms of post-processing for a 10.4 ms network at batch 16 - with no detection at all.)"""
    coder = BoxCoder((1.0, 1.0, 1.0, 1.0))
    num_images = cls_logits_per_level[0].shape[0]
    dev = cls_logits_per_level[0].device
    # sigmoid is monotone: select on the (tf-idf scaled) logits with the threshold mapped to logit space
    thr_logit = math.log(score_thresh / (1.0 - score_thresh)) if 0.0 < score_thresh < 1.0 else float("-inf")
    if _RETINA_FUSED and len(cls_logits_per_level) <= 8 and topk_candidates <= 16384:
        # the whole chain below as ONE host call (`mi355det_retina_detections`: same kernels for top-k and NMS, one kernel for the index
        # arithmetic / decode / clip of all levels, one for the final gather) and one read
        lg = cls_logits_per_level if tfidf_post is None else [l * tfidf_post for l in cls_logits_per_level]
        lim = _clip_limits(image_shapes, dev, torch.float32).reshape(num_images, 4)
        b, sc, l, counts = ops.retina_detections(lg, bbox_reg_per_level, anchors_per_level, lim, thr_logit, topk_candidates, nms_thresh,
                                                 detections_per_img, coder.bbox_xform_clip)
        counts = counts.tolist()                                                              # the one synchronisation
        return [{"boxes": b[i, :c], "scores": sc[i, :c], "labels": l[i, :c]} for i, c in enumerate(counts)]
    batch = torch.arange(num_images, device=dev)[:, None]
    lb, ls, ll, lv = [], [], [], []
    for logits, reg, anchors in zip(cls_logits_per_level, bbox_reg_per_level, anchors_per_level):
        lg = logits if tfidf_post is None else logits * tfidf_post
        K = lg.shape[-1]
        flat = lg.reshape(num_images, -1)
        k = min(topk_candidates, flat.shape[1])
        val, idx, cnt = ops.topk_rows(flat, k, min_value=thr_logit)                       # every image of the level at once
        valid = torch.arange(k, device=dev)[None, :] < cnt[:, None]
        a_idx, lab = idx // K, idx % K
        bx = coder.decode_single(reg[batch, a_idx].reshape(-1, 4), anchors[a_idx].reshape(-1, 4)).reshape(num_images, k, 4)
        lb.append(bx)
        ls.append(torch.sigmoid(val))
        ll.append(lab)
        lv.append(valid)
    b, sc, l, valid = torch.cat(lb, 1), torch.cat(ls, 1), torch.cat(ll, 1), torch.cat(lv, 1)
    lim = _clip_limits(image_shapes, dev, b.dtype)
    b = torch.minimum(b.clamp(min=0), lim)                                                # clip_boxes_to_image
    b = torch.where(valid[..., None], b, torch.zeros_like(b))
    masked = torch.where(valid, sc, torch.full_like(sc, float("-inf")))
    keeps, cnt = ops.nms_batch(b, masked, nms_thresh, idxs=l)
    n_all = b.shape[1]
    good = (torch.arange(n_all, device=dev)[None, :] < cnt[:, None]) & torch.gather(valid, 1, keeps.clamp(max=n_all - 1))
    counts = good.sum(1).clamp(max=detections_per_img).tolist()                           # the one synchronisation
    detections = []
    for i in range(num_images):
        kp = keeps[i][:counts[i]]
        detections.append({"boxes": b[i][kp], "scores": sc[i][kp], "labels": l[i][kp]})
    return detections


def roi_heads_postprocess_detections(class_logits, box_regression, proposals, image_shapes, tfidf_post=1.0, score_thresh=0.05,
                                     nms_thresh=0.5, detections_per_img=100, weights=(10.0, 10.0, 5.0, 5.0), loss_type="ce"):
    """RoIHeads.postprocess_detections (roi_heads.py:715-781): scores by the training loss ('ce' softmax, 'gombit*' double-exponential,
    otherwise sigmoid; :724-729), per-class batched NMS."""
    coder = BoxCoder(weights)
    num_classes = class_logits.shape[-1]
    per_img = [len(p) for p in proposals]
    pred_boxes = coder.decode(box_regression, proposals)
    if loss_type == "ce":
        pred_scores = torch.softmax(tfidf_post * class_logits, -1)
    elif loss_type.startswith("gombit"):
        pred_scores = 1 / (torch.exp(torch.exp(-tfidf_post * (class_logits - 1.96))))
    else:
        pred_scores = torch.sigmoid(tfidf_post * class_logits)
    out_b, out_s, out_l = [], [], []
    for boxes, scores, shape in zip(pred_boxes.split(per_img, 0), pred_scores.split(per_img, 0), image_shapes):
        boxes = box_ops.clip_boxes_to_image(boxes, shape)
        labels = torch.arange(num_classes, device=boxes.device).view(1, -1).expand_as(scores)
        boxes, scores, labels = boxes[:, 1:].reshape(-1, 4), scores[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)
        inds = torch.nonzero(scores > score_thresh).squeeze(1)
        boxes, scores, labels = boxes[inds], scores[inds], labels[inds]
        keep = box_ops.remove_small_boxes(boxes, 1e-2)
        boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
        kb, ks, kl = batched_nms_topn(boxes, scores, labels, nms_thresh, detections_per_img)
        out_b.append(kb)
        out_s.append(ks)
        out_l.append(kl)
    return out_b, out_s, out_l


ROI_DET_MAX_CANDIDATES = 4096      # (proposal, class) pairs above the score threshold per image the one-call route handles


def roi_heads_postprocess_detections_batch(class_logits, box_regression, proposals_pad, proposal_counts, image_shapes, tfidf_post=1.0,
                                           score_thresh=0.05, nms_thresh=0.5, detections_per_img=100, weights=(10.0, 10.0, 5.0, 5.0),
                                           loss_type="ce", max_candidates=ROI_DET_MAX_CANDIDATES):
    """RoIHeads.postprocess_detections (roi_heads.py:715-781) on PADDED proposals [N, P, 4] (+ their counts [N] int32 on the device) with one
    host call (`mi355det_roi_detections`) and ONE host read: class_logits [N*P, C], box_regression [N*P, C*4] in proposal order.
    -> (boxes, scores, labels) lists like roi_heads_postprocess_detections, or None when an image has max_candidates or more scores above
    the threshold (the caller then takes the unbounded per-image route)."""
    n, p = proposals_pad.shape[0], proposals_pad.shape[1]
    c = class_logits.shape[-1]
    if loss_type == "ce":
        scores = torch.softmax(tfidf_post * class_logits, -1)
    elif loss_type.startswith("gombit"):
        scores = 1 / (torch.exp(torch.exp(-tfidf_post * (class_logits - 1.96))))
    else:
        scores = torch.sigmoid(tfidf_post * class_logits)
    scores = scores.reshape(n, p, c)
    scores[:, :, 0] = float("-inf")                                     # the background column is dropped (roi_heads.py:744-747)
    pad = torch.arange(p, device=scores.device)[None, :] >= proposal_counts[:, None]
    scores.masked_fill_(pad[:, :, None], float("-inf"))                 # rows of the padding
    lim = _clip_limits(image_shapes, scores.device, torch.float32).reshape(n, 4)
    boxes, out_s, labels, meta, k = ops.roi_detections(scores, box_regression, proposals_pad, lim, score_thresh, max_candidates, weights, nms_thresh,
                                                       detections_per_img)
    host = meta.tolist()                                                # the one synchronisation of the inference step
    if any(host[n + i] >= k for i in range(n)):
        return None
    return ([boxes[i, :host[i]] for i in range(n)], [out_s[i, :host[i]] for i in range(n)], [labels[i, :host[i]] for i in range(n)])


NMS_CAPACITY = 16384     # boxes per launch of the NMS kernels (include/mi355det.h)


def batched_nms_topn(boxes, scores, labels, nms_thresh, top_n, capacity=NMS_CAPACITY):
    """boxes[keep], scores[keep], labels[keep] for keep = batched_nms(boxes, scores, labels, nms_thresh)[:top_n] (roi_heads.py:771-774), for ANY
    number of candidates (LVIS: 1000 proposals x 1203 classes).  Greedy NMS only ever lets a HIGHER-scoring kept box suppress a lower one, so
    the candidates can be consumed in descending-score chunks: the survivors of the best `capacity` candidates are final, and if there are at
    least top_n of them the rest cannot enter the result.  Otherwise the next chunk is judged together with ALL survivors so far (fewer than
    top_n, so they fit beside it).  Exact, not a truncation."""
    n = boxes.shape[0]
    if top_n >= capacity:
        raise ValueError(f"batched_nms_topn: top_n must be below the NMS capacity ({capacity})")
    if n <= capacity:
        keep = box_ops.batched_nms(boxes, scores, labels, nms_thresh)[:top_n]
        return boxes[keep], scores[keep], labels[keep]
    # the per-category coordinate offsets of batched_nms depend on the largest coordinate of the set it is given: fix them once for the
    # whole candidate set, as a single call over all candidates would
    off = labels.to(boxes) * (boxes.max() + 1)
    shifted = boxes + off[:, None]
    remaining = scores.clone()
    kept_idx = torch.empty((0,), dtype=torch.int64, device=boxes.device)
    taken = 0
    while taken < n:
        room = capacity - kept_idx.numel()
        k = min(room, n - taken)
        _v, idx, _c = ops.topk_rows(remaining.reshape(1, -1), k)      # next k best (descending score, ties by index)
        chunk = idx[0]
        remaining[chunk] = float("-inf")
        taken += k
        cand = torch.cat([kept_idx, chunk])                            # survivors first: they outrank the chunk and never suppress each other
        keep = ops.nms(shifted[cand], scores[cand], nms_thresh)
        kept_idx = cand[keep]
        if kept_idx.numel() >= top_n:
            break
    kept_idx = kept_idx[:top_n]
    return boxes[kept_idx], scores[kept_idx], labels[kept_idx]
