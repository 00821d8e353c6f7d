"""Training-side mirrors of `RegionProposalNetwork` (tvision/rpn.py) over the HIP kernels.

  assign_targets_to_anchors   rpn.py:179-213   fused box_iou + Matcher(0.7, 0.3, allow_low_quality) per image, no [M,N] matrix
  compute_loss                rpn.py:282-318   sampler (torch, row a19) + smooth-L1 / BCE on the <= 256 sampled anchors per image
  filter_proposals            rpn.py:215-280   -> tvision/postprocess.py:rpn_filter_proposals
"""
import torch
import torch.nn.functional as F

from ._utils import BalancedPositiveNegativeSampler, BoxCoder, Matcher


class RPNTargets:
    """The target-side state of the reference RPN: matcher, sampler, coder with the reference's defaults (frcnn.py:170-190)."""

    def __init__(self, fg_iou_thresh=0.7, bg_iou_thresh=0.3, batch_size_per_image=256, positive_fraction=0.5):
        self.proposal_matcher = Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=True)
        self.fg_bg_sampler = BalancedPositiveNegativeSampler(batch_size_per_image, positive_fraction)
        self.box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))

    def assign_targets_to_anchors(self, anchors, targets):
        labels, matched_gt_boxes = [], []
        for anchors_per_image, t in zip(anchors, targets):
            gt = t["boxes"]
            if gt.numel() == 0:       # background image (rpn.py:186-190)
                matched_gt_boxes.append(torch.zeros_like(anchors_per_image, dtype=torch.float32))
                labels.append(torch.zeros((anchors_per_image.shape[0],), dtype=torch.float32, device=anchors_per_image.device))
                continue
            m = self.proposal_matcher.match_boxes(gt, anchors_per_image)
            matched_gt_boxes.append(gt[m.clamp(min=0)])
            lab = (m >= 0).to(torch.float32)
            lab[m == Matcher.BELOW_LOW_THRESHOLD] = 0.0
            lab[m == Matcher.BETWEEN_THRESHOLDS] = -1.0
            labels.append(lab)
        return labels, matched_gt_boxes

    def compute_loss(self, objectness, pred_bbox_deltas, labels, regression_targets):
        pos, neg = self.fg_bg_sampler(labels)
        pos = torch.where(torch.cat(pos, dim=0))[0]
        neg = torch.where(torch.cat(neg, dim=0))[0]
        sampled = torch.cat([pos, neg], dim=0)
        objectness = objectness.flatten()
        labels = torch.cat(labels, dim=0)
        regression_targets = torch.cat(regression_targets, dim=0)
        box_loss = F.smooth_l1_loss(pred_bbox_deltas[pos], regression_targets[pos], beta=1 / 9, reduction="sum") / sampled.numel()
        objectness_loss = F.binary_cross_entropy_with_logits(objectness[sampled], labels[sampled])
        return objectness_loss, box_loss

    def prepare(self, anchors, targets):
        """Everything of the RPN loss that does not depend on the network outputs: labels, regression targets, sampled indices."""
        labels, matched = self.assign_targets_to_anchors(anchors, targets)
        reg = self.box_coder.encode(matched, anchors)
        pos, neg = self.fg_bg_sampler(labels)
        sizes = getattr(self.fg_bg_sampler, "last_counts", None)
        if sizes is not None and len(sizes) == len(pos):      # the sampler knows how many it drew: index lists without a read-back
            pos = torch.nonzero_static(torch.cat(pos, dim=0), size=sum(a for a, _ in sizes)).squeeze(1)
            neg = torch.nonzero_static(torch.cat(neg, dim=0), size=sum(b for _, b in sizes)).squeeze(1)
        else:
            pos = torch.where(torch.cat(pos, dim=0))[0]
            neg = torch.where(torch.cat(neg, dim=0))[0]
        return dict(pos=pos, sampled=torch.cat([pos, neg], dim=0), labels=torch.cat(labels, dim=0), reg=torch.cat(reg, dim=0))

    def losses_prepared(self, objectness, pred_bbox_deltas, prep):
        """compute_loss (rpn.py:282-318) on the indices / targets of prepare()."""
        pos, sampled = prep["pos"], prep["sampled"]
        box_loss = F.smooth_l1_loss(pred_bbox_deltas[pos], prep["reg"][pos], beta=1 / 9, reduction="sum") / sampled.numel()
        objectness_loss = F.binary_cross_entropy_with_logits(objectness.flatten()[sampled], prep["labels"][sampled])
        return {"loss_objectness": objectness_loss, "loss_rpn_box_reg": box_loss}

    def losses_prepared_fused(self, objectness, pred_bbox_deltas, prep):
        """losses_prepared + its gradients in one launch (`mi355det_rpn_loss`), outside autograd:
        -> ({'loss_objectness', 'loss_rpn_box_reg'}, d loss / d objectness, d loss / d pred_bbox_deltas) for a unit weight on both losses."""
        from .. import ops
        losses, g_obj, g_dl = ops.rpn_loss(objectness.detach(), pred_bbox_deltas.detach(), prep["labels"], prep["reg"], prep["pos"], prep["sampled"])
        return {"loss_objectness": losses[0], "loss_rpn_box_reg": losses[1]}, g_obj, g_dl

    def losses(self, objectness, pred_bbox_deltas, anchors, targets):
        """rpn.py:353-361: -> {'loss_objectness', 'loss_rpn_box_reg'}."""
        labels, matched = self.assign_targets_to_anchors(anchors, targets)
        reg = self.box_coder.encode(matched, anchors)
        lo, lb = self.compute_loss(objectness, pred_bbox_deltas, labels, reg)
        return {"loss_objectness": lo, "loss_rpn_box_reg": lb}
