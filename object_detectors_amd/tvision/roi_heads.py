"""Training-side mirrors of `RoIHeads` (tvision/roi_heads.py) over the HIP kernels.

  fastrcnn_loss                 roi_heads.py:22-98     'ce' | 'bce' | 'focal_loss' (the gombit variants are not on this path)
  assign_targets_to_proposals   roi_heads.py:627-651   fused box_iou + Matcher(0.5, 0.5) per image
  select_training_samples       roi_heads.py:688-713   + add_gt_proposals, sampler (torch, row a19), BoxCoder(10,10,5,5).encode
  postprocess_detections        roi_heads.py:715-781   -> tvision/postprocess.py:roi_heads_postprocess_detections
  box_roi_pool                  roi_heads.py:818       -> tvision/roi_align.py:MultiScaleRoIAlign
"""
import torch
import torch.nn.functional as F

from ._utils import BalancedPositiveNegativeSampler, BoxCoder, Matcher
from .focal_loss import sigmoid_focal_loss


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets, weights=None, loss_type="ce"):
    labels = torch.cat(labels, dim=0)
    bs = labels.shape[0]
    regression_targets = torch.cat(regression_targets, dim=0)
    if loss_type == "ce":
        classification_loss = F.cross_entropy(class_logits, labels, weight=weights)
    else:
        y = torch.zeros_like(class_logits)
        y.scatter_(1, labels.unsqueeze(1), 1)
        y[:, 0] = 0.0
        if loss_type == "bce":
            classification_loss = F.binary_cross_entropy_with_logits(class_logits, y, reduction="sum") / bs
        elif loss_type == "focal_loss":
            classification_loss = sigmoid_focal_loss(class_logits, y, reduction="sum") / bs
        else:
            raise NotImplementedError(f"loss_type {loss_type!r}: only 'ce', 'bce', 'focal_loss' are on the accelerated path")
    pos = torch.nonzero(labels > 0).squeeze(1)
    labels_pos = labels[pos]
    n = class_logits.shape[0]
    box_regression = box_regression.reshape(n, -1, 4)
    box_loss = F.smooth_l1_loss(box_regression[pos, labels_pos], regression_targets[pos], reduction="sum") / labels.numel()
    return classification_loss, box_loss


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
        return [torch.nonzero(p | n).squeeze(1) for p, n in zip(pos, neg)]

    @staticmethod
    def add_gt_proposals(proposals, gt_boxes):
        return [torch.cat((p, g)) for p, g in zip(proposals, gt_boxes)]

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
