"""Mirror of torchvision.ops.roi_align / MultiScaleRoIAlign as the reference uses them
(tvision/frcnn.py:208-211 `MultiScaleRoIAlign(['0','1','2','3'], 7, 2)`, tvision/roi_heads.py:818)."""
import math

import torch
import torch.nn as nn

from .. import ops


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, output_size, scales, sampling_ratio, aligned, k_min, k_max, *feats):
        ctx.args = (output_size, scales, sampling_ratio, aligned, k_min, k_max)
        ctx.save_for_backward(rois, *feats)
        return ops.roi_align_multi(list(feats), rois, output_size, scales, sampling_ratio, aligned, k_min, k_max)

    @staticmethod
    def backward(ctx, g):
        rois, *feats = ctx.saved_tensors
        output_size, scales, sampling_ratio, aligned, k_min, k_max = ctx.args
        dfs = ops.roi_align_multi(list(feats), rois, output_size, scales, sampling_ratio, aligned, k_min, k_max, grad_out=g)
        return (None,) * 7 + tuple(dfs)


class _RoIAlignNHWCFn(torch.autograd.Function):
    """bf16 NHWC features (the engines' layout) -> fp32 [K,C,ph,pw]; feature gradients come back as bf16 NHWC."""

    @staticmethod
    def forward(ctx, rois, output_size, scales, sampling_ratio, aligned, k_min, k_max, *feats):
        ctx.args = (output_size, scales, sampling_ratio, aligned, k_min, k_max)
        ctx.save_for_backward(rois, *feats)
        return ops.roi_align_nhwc(list(feats), rois, output_size, scales, sampling_ratio, aligned, k_min, k_max)

    @staticmethod
    def backward(ctx, g):
        rois, *feats = ctx.saved_tensors
        output_size, scales, sampling_ratio, aligned, k_min, k_max = ctx.args
        dfs = ops.roi_align_nhwc(list(feats), rois, output_size, scales, sampling_ratio, aligned, k_min, k_max, grad_out=g)
        return (None,) * 7 + tuple(d.to(torch.bfloat16) for d in dfs)


def _rois_tensor(boxes):
    if isinstance(boxes, torch.Tensor):
        return boxes
    ids = torch.cat([torch.full((b.shape[0], 1), i, dtype=b.dtype, device=b.device) for i, b in enumerate(boxes)])
    return torch.cat([ids, torch.cat(list(boxes))], dim=1)


def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    return _RoIAlignFn.apply(_rois_tensor(boxes), output_size, [spatial_scale], sampling_ratio, aligned, 0, 0, input)


class MultiScaleRoIAlign(nn.Module):
    def __init__(self, featmap_names, output_size, sampling_ratio, canonical_scale=224, canonical_level=4):
        super().__init__()
        self.featmap_names = featmap_names
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.sampling_ratio = sampling_ratio
        if canonical_scale != 224 or canonical_level != 4:
            raise NotImplementedError("LevelMapper constants are those of the reference (224, 4)")

    def forward(self, x, boxes, image_shapes):
        feats = [v for k, v in x.items() if k in self.featmap_names]
        if len(feats) > 4:
            raise ValueError("at most 4 pyramid levels")
        max_h = max(s[0] for s in image_shapes)
        max_w = max(s[1] for s in image_shapes)
        scales = []
        for f in feats:   # infer_scale: 2 ** round(log2(feat/img))
            sh = 2.0 ** round(math.log2(f.shape[-2] / max_h))
            sw = 2.0 ** round(math.log2(f.shape[-1] / max_w))
            assert sh == sw
            scales.append(sh)
        k_min, k_max = int(-math.log2(scales[0])), int(-math.log2(scales[-1]))
        return _RoIAlignFn.apply(_rois_tensor(boxes), self.output_size, scales, self.sampling_ratio, False, k_min, k_max, *feats)

    def forward_nhwc(self, feats, boxes, image_shapes):
        """Same pooling on bf16 NHWC feature maps [n,h,w,C] (list, finest first): no layout conversion, coalesced backward."""
        if len(feats) > 4:
            raise ValueError("at most 4 pyramid levels")
        max_h = max(s[0] for s in image_shapes)
        max_w = max(s[1] for s in image_shapes)
        scales = []
        for f in feats:
            sh = 2.0 ** round(math.log2(f.shape[1] / max_h))
            sw = 2.0 ** round(math.log2(f.shape[2] / max_w))
            assert sh == sw
            scales.append(sh)
        k_min, k_max = int(-math.log2(scales[0])), int(-math.log2(scales[-1]))
        return _RoIAlignNHWCFn.apply(_rois_tensor(boxes), self.output_size, scales, self.sampling_ratio, False, k_min, k_max, *feats)
