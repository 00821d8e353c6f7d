"""Mirror of torchvision_models/tvision/_utils.py: Matcher and BoxCoder over HIP kernels."""
import math

import torch

from .. import ops


class BoxCoder(object):
    """tvision/_utils.py:128-223."""

    def __init__(self, weights, bbox_xform_clip=math.log(1000. / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def encode(self, reference_boxes, proposals):
        boxes_per_image = [len(b) for b in reference_boxes]
        targets = self.encode_single(torch.cat(reference_boxes, dim=0), torch.cat(proposals, dim=0))
        return targets.split(boxes_per_image, 0)

    def encode_single(self, reference_boxes, proposals):
        if proposals.shape[0] == 0:
            return proposals.new_zeros((0, 4))
        return ops.box_encode(reference_boxes, proposals, self.weights)

    def decode(self, rel_codes, boxes):
        assert isinstance(boxes, (list, tuple))
        concat = torch.cat(boxes, dim=0)
        box_sum = concat.shape[0]
        if box_sum > 0:
            rel_codes = rel_codes.reshape(box_sum, -1)
        pred = self.decode_single(rel_codes, concat)
        if box_sum > 0:
            pred = pred.reshape(box_sum, -1, 4)
        return pred

    def decode_single(self, rel_codes, boxes):
        if boxes.shape[0] == 0:
            return rel_codes.new_zeros(rel_codes.shape)
        return ops.box_decode(rel_codes, boxes, self.weights, self.bbox_xform_clip)


class Matcher(object):
    """tvision/_utils.py:226-344.  `__call__` keeps the reference's [M,N] quality-matrix signature;
    `match_boxes` is the fused form (box_iou + matching, no [M,N] matrix) used by the model mirrors."""
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        assert low_threshold <= high_threshold
        self.high_threshold = high_threshold
        self.low_threshold = low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def match_boxes(self, gt_boxes, anchors):
        return ops.match_anchors(gt_boxes, anchors, self.high_threshold, self.low_threshold, self.allow_low_quality_matches)

    def __call__(self, match_quality_matrix):
        q = match_quality_matrix
        if q.numel() == 0:
            if q.shape[0] == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            raise ValueError("No proposal boxes available for one of the images during training")
        # a materialised matrix was handed over: thresholds on it (same semantics as the fused kernel)
        vals, matches = q.max(dim=0)
        allm = matches.clone()
        matches[vals < self.low_threshold] = self.BELOW_LOW_THRESHOLD
        matches[(vals >= self.low_threshold) & (vals < self.high_threshold)] = self.BETWEEN_THRESHOLDS
        if self.allow_low_quality_matches:
            best, _ = q.max(dim=1)
            pred = torch.where(q == best[:, None])[1]
            matches[pred] = allm[pred]
        return matches


class BalancedPositiveNegativeSampler(object):
    """tvision/_utils.py:9-76.  Stays in torch by design: it is RNG-dependent (`torch.randperm`), tiny
    (<= 512 indices per image) and must consume the framework's generator to stay reproducible against the reference."""

    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction

    def __call__(self, matched_idxs):
        """Same draws as the reference loop (`torch.where` + `torch.randperm(count)` per image, positives then negatives, in image order:
        _utils.py:38-73), but with ONE device-to-host read for the whole batch instead of three per image: the positive / negative counts of
        all images are read together, `randperm` is then called with exactly the sizes and in exactly the order of the reference, and the
        index lists come from `nonzero_static` (ascending, like `where`), whose size is already known."""
        pos_masks = [m >= 1 for m in matched_idxs]
        neg_masks = [m == 0 for m in matched_idxs]
        if not matched_idxs:
            return [], []
        counts = torch.stack([pm.sum() for pm in pos_masks] + [nm.sum() for nm in neg_masks]).tolist()
        n_img = len(matched_idxs)
        pos_idx, neg_idx = [], []
        for i, m in enumerate(matched_idxs):
            cp, cn = int(counts[i]), int(counts[n_img + i])
            positive = torch.nonzero_static(pos_masks[i], size=cp).squeeze(1)
            negative = torch.nonzero_static(neg_masks[i], size=cn).squeeze(1)
            num_pos = min(cp, int(self.batch_size_per_image * self.positive_fraction))
            num_neg = min(cn, self.batch_size_per_image - num_pos)
            perm1 = torch.randperm(cp, device=m.device)[:num_pos]
            perm2 = torch.randperm(cn, device=m.device)[:num_neg]
            pm = torch.zeros_like(m, dtype=torch.uint8)
            nm = torch.zeros_like(m, dtype=torch.uint8)
            pm[positive[perm1]] = 1
            nm[negative[perm2]] = 1
            pos_idx.append(pm)
            neg_idx.append(nm)
        self.last_counts = [(min(int(counts[i]), int(self.batch_size_per_image * self.positive_fraction)),
                             min(int(counts[n_img + i]), self.batch_size_per_image - min(int(counts[i]), int(self.batch_size_per_image * self.positive_fraction))))
                            for i in range(n_img)]
        return pos_idx, neg_idx


def smooth_l1_loss(input, target, beta: float = 1. / 9, size_average: bool = True):
    """tvision/_utils.py:347-358."""
    n = torch.abs(input - target)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    return loss.mean() if size_average else loss.sum()
