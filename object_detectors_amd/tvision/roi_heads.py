"""Training-side mirrors of `RoIHeads` (tvision/roi_heads.py) over the HIP kernels.

  fastrcnn_loss                 roi_heads.py:22-98     'ce' | 'bce' | 'focal_loss' | 'gombit' | 'gombit_fl', fused forward + gradients (mi355det_fastrcnn_loss)
  minibatch_tfidf               roi_heads.py:801-809   per-batch smoothed idf row
  assign_targets_to_proposals   roi_heads.py:627-651   fused box_iou + Matcher(0.5, 0.5) per image
  select_training_samples       roi_heads.py:688-713   + add_gt_proposals, sampler (torch, row a19), BoxCoder(10,10,5,5).encode
  postprocess_detections        roi_heads.py:715-781   -> tvision/postprocess.py:roi_heads_postprocess_detections
  box_roi_pool                  roi_heads.py:818       -> tvision/roi_align.py:MultiScaleRoIAlign
"""
import torch

from .. import ops

from ._utils import BalancedPositiveNegativeSampler, BoxCoder, Matcher


class _FastRCNNLossFn(torch.autograd.Function):
    """Forward and both gradients come out of ONE kernel launch sequence (csrc/frcnn_kernels.hip); backward only scales them."""

    @staticmethod
    def forward(ctx, class_logits, box_regression, labels, regression_targets, class_scale, class_weights, loss_type):
        losses, gl, gb = ops.fastrcnn_loss(class_logits, box_regression, labels, regression_targets, class_scale, class_weights, loss_type, want_grad=True)
        ctx.save_for_backward(gl, gb)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_cls, g_box):
        gl, gb = ctx.saved_tensors
        return gl * g_cls, gb * g_box, None, None, None, None, None


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets, weights=None, loss_type="ce", class_scale=None):
    """roi_heads.py:24-96: (classification_loss, box_loss) for loss_type 'ce' | 'bce' | 'focal_loss' | 'gombit' | 'gombit_fl'.

    The reference multiplies the tf-idf row into the logits at the call site (:826-827 `fastrcnn_loss(self.tfidf * class_logits, ...)`);
    pass that row as `class_scale` to have the kernel apply it (the returned gradient is w.r.t. the unscaled logits), or pre-multiply and
    leave it None, as the reference's callers do."""
    labels = torch.cat(labels, dim=0)
    regression_targets = torch.cat(regression_targets, dim=0)
    if loss_type not in ops.FRCNN_LOSS_TYPES:
        raise ValueError(f"fastrcnn_loss: unknown loss_type {loss_type!r}")
    if weights is not None and loss_type != "ce":
        weights = None                     # the reference passes `weights` along but only F.cross_entropy reads it (roi_heads.py:45-46)
    return _FastRCNNLossFn.apply(class_logits, box_regression, labels, regression_targets, class_scale, weights, loss_type)


def minibatch_tfidf(targets, num_classes, tfidf_norm=0):
    """roi_heads.py:801-809: smoothed idf of the classes present in this mini-batch, optionally p-normalised ([num_classes] float)."""
    w = torch.stack([torch.bincount(t["labels"], minlength=num_classes) for t in targets])
    w[w > 0] = 1
    w = w.sum(axis=0)
    w = torch.log((len(targets) + 1) / (w + 1)) + 1
    if tfidf_norm != 0:
        w = w / torch.norm(w, p=tfidf_norm)
    return w


class RoIHeadTargets:
    """Target-side state of the reference RoIHeads with its defaults (frcnn.py:226-236)."""

    def __init__(self, fg_iou_thresh=0.5, bg_iou_thresh=0.5, batch_size_per_image=512, positive_fraction=0.25, bbox_reg_weights=(10., 10., 5., 5.)):
        self.proposal_matcher = Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=False)
        self.fg_bg_sampler = BalancedPositiveNegativeSampler(batch_size_per_image, positive_fraction)
        self.box_coder = BoxCoder(bbox_reg_weights)

    def assign_targets_to_proposals(self, proposals, gt_boxes, gt_labels):
        matched_idxs, labels = [], []
        for p, gb, gl in zip(proposals, gt_boxes, gt_labels):
            m = self.proposal_matcher.match_boxes(gb, p)
            clamped = m.clamp(min=0)
            lab = gl[clamped].to(torch.int64)
            lab[m == Matcher.BELOW_LOW_THRESHOLD] = 0
            lab[m == Matcher.BETWEEN_THRESHOLDS] = -1
            matched_idxs.append(clamped)
            labels.append(lab)
        return matched_idxs, labels

    def subsample(self, labels):
        pos, neg = self.fg_bg_sampler(labels)
        sizes = getattr(self.fg_bg_sampler, "last_counts", None)
        if sizes is not None and len(sizes) == len(pos):
            # the sampler already knows how many it drew per image: no read-back for the index lists (roi_heads.py:654-662)
            return [torch.nonzero_static(p | n, size=a + b).squeeze(1) for p, n, (a, b) in zip(pos, neg, sizes)]
        return [torch.nonzero(p | n).squeeze(1) for p, n in zip(pos, neg)]

    @staticmethod
    def add_gt_proposals(proposals, gt_boxes):
        return [torch.cat((p, g)) for p, g in zip(proposals, gt_boxes)]

    # limits of mi355det_roi_match / mi355det_roi_sample (include/mi355det.h); beyond them the composed route below is used
    FUSED_MAX_IMAGES, FUSED_MAX_GT, FUSED_MAX_CANDIDATES, FUSED_MAX_SAMPLES = 64, 1024, 8192, 1024

    def fused_ok(self, n_images, max_proposals, targets):
        g = [int(t["boxes"].shape[0]) for t in targets]
        return (n_images <= self.FUSED_MAX_IMAGES and min(g) >= 1 and max(g) <= self.FUSED_MAX_GT
                and max_proposals + max(g) <= self.FUSED_MAX_CANDIDATES and self.fg_bg_sampler.batch_size_per_image <= self.FUSED_MAX_SAMPLES)

    def select_training_samples_fused(self, proposals_pad, meta, targets):
        """select_training_samples (roi_heads.py:664-713) on the padded proposals [N, P, 4] of `ops.rpn_proposals`, whose counts sit in
        meta[:N] (int32 [3N] on the device; meta[N:] receives the positive / negative counts): two launches around ONE host read - which
        is also the read of the proposal counts.  `torch.randperm` is called with the reference's sizes in the reference's order
        (_utils.py:38-73), so the samples are those of the composed route under the same generator state (tests/test_gpu_proposals.py).
        -> (rois [S,5], matched_idxs [S], labels [S], regression_targets [S,4], samples per image)."""
        n = proposals_pad.shape[0]
        dev = proposals_pad.device
        gt_boxes = [t["boxes"].to(torch.float32) for t in targets]
        offs = [0]
        for g in gt_boxes:
            offs.append(offs[-1] + int(g.shape[0]))
        gt_all = torch.cat(gt_boxes)
        gl_all = torch.cat([t["labels"] for t in targets])
        m = self.proposal_matcher
        matched, labels, _ = ops.roi_match(proposals_pad, meta[:n], gt_all, gl_all, offs, m.high_threshold, m.low_threshold,
                                           counts_out=meta[n:].view(n, 2))
        host = meta.tolist()                                  # the one synchronisation of proposals + sampling
        bs, frac = self.fg_bg_sampler.batch_size_per_image, self.fg_bg_sampler.positive_fraction
        perm_pos, perm_neg, num_pos, num_neg = [], [], [], []
        for i in range(n):
            cp, cn = host[n + 2 * i], host[n + 2 * i + 1]
            npos = min(cp, int(bs * frac))
            num_pos.append(npos)
            num_neg.append(min(cn, bs - npos))
            perm_pos.append(torch.randperm(cp, device=dev))
            perm_neg.append(torch.randperm(cn, device=dev))
        rois, out_l, out_m, reg = ops.roi_sample(proposals_pad, meta[:n], gt_all, offs, matched, labels, perm_pos, perm_neg, num_pos, num_neg,
                                                 self.box_coder.weights)
        return rois, out_m, out_l, reg, [a + b for a, b in zip(num_pos, num_neg)]

    def select_training_samples(self, proposals, targets):
        assert targets is not None and all("boxes" in t and "labels" in t for t in targets)
        dtype = proposals[0].dtype
        gt_boxes = [t["boxes"].to(dtype) for t in targets]
        gt_labels = [t["labels"] for t in targets]
        proposals = self.add_gt_proposals(proposals, gt_boxes)
        matched_idxs, labels = self.assign_targets_to_proposals(proposals, gt_boxes, gt_labels)
        sampled = self.subsample(labels)
        matched_gt_boxes = []
        for i in range(len(proposals)):
            proposals[i] = proposals[i][sampled[i]]
            labels[i] = labels[i][sampled[i]]
            matched_idxs[i] = matched_idxs[i][sampled[i]]
            matched_gt_boxes.append(gt_boxes[i][matched_idxs[i]])
        regression_targets = self.box_coder.encode(matched_gt_boxes, proposals)
        return proposals, matched_idxs, labels, regression_targets
