"""Mirror of the reference `RetinaNet` / `retinanet_resnet50_fpn` (tvision/retinanet.py:249-660) over the MI355X engine.

    model = retinanet_resnet50_fpn(num_classes=91, tfidf={...})      # same call as the reference's detection/train.py
    losses = model(images, targets)        # training: {'classification': ..., 'bbox_regression': ...} (backward already done)
    detections = model(images)             # eval: [{'boxes','scores','labels'}]

Inputs: a LIST of [3,H,W] images of any sizes goes through the reference's GeneralizedRCNNTransform (normalise, bilinear resize to
min_size 800 / max_size 1333, zero-pad to a multiple of 32, targets' boxes rescaled; detections mapped back to the input frame) on the
GPU (tvision/transform.py); a ready [N,3,H,W] batch (H, W multiples of 32) skips resize / padding and has the normalisation fused into
the stem's im2col.  In training mode the call runs the fused forward + loss + backward and leaves the gradients in
`model.engine.flat_g` (use `object_detectors_amd.optim.FlatSGD.for_engine(model.engine)`), returning detached losses.
"""
import torch
from torch import nn

from .engine import IMAGE_MEAN, IMAGE_STD, RetinaNetEngine
from .postprocess import retinanet_postprocess_detections
from .transform import GeneralizedRCNNTransform


class RetinaNet(nn.Module):
    def __init__(self, num_classes=91, trainable_backbone_layers=3, score_thresh=0.05, nms_thresh=0.5, detections_per_img=300,
                 topk_candidates=1000, tfidf=None, device=None, seed=0, body="resnet50", min_size=800, max_size=1333, image_mean=None,
                 image_std=None):
        super().__init__()
        self.engine = RetinaNetEngine(num_classes, 9, trainable_backbone_layers, device=device, seed=seed, body=body)
        self.transform = GeneralizedRCNNTransform(min_size, max_size, image_mean or list(IMAGE_MEAN), image_std or list(IMAGE_STD))   # retinanet.py:383
        self.score_thresh, self.nms_thresh = score_thresh, nms_thresh
        self.detections_per_img, self.topk_candidates = detections_per_img, topk_candidates
        self.tfidf = None if tfidf is None else tfidf["values"].to(self.engine.device).float()
        self.tfidf_post = self.tfidf

    def state_dict(self, *a, **k):
        return self.engine.reference_state_dict()

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.engine.load_reference_state_dict(sd)

    def forward(self, images, targets=None):
        if self.training:
            if targets is None:
                raise ValueError("In training mode, targets should be passed")          # retinanet.py:489-490
            for t in targets:
                b = t["boxes"]
                if b.dim() != 2 or b.shape[-1] != 4:
                    raise ValueError("Expected target boxes to be a tensor of shape [N, 4], got {:}.".format(b.shape))   # retinanet.py:495-501
        original_image_sizes = None
        if isinstance(images, (list, tuple)):
            original_image_sizes = [(int(i.shape[-2]), int(i.shape[-1])) for i in images]                                # retinanet.py:505-509
            self.transform.train(self.training)
            image_list, targets = self.transform(images, targets)                                                        # retinanet.py:512
            images, shapes = image_list.tensors, image_list.image_sizes
            self.engine.normalize = False              # the transform normalised BEFORE padding: padded pixels are exact zeros
        else:
            shapes = [(images.shape[-2], images.shape[-1])] * images.shape[0]
            self.engine.normalize = True
        if self.training:
            for t in targets:
                b = t["boxes"]
                if b.numel() and bool(((b[:, 2:] <= b[:, :2]).any())):
                    raise ValueError("All bounding boxes should have positive height and width.")                        # retinanet.py:515-525
            losses = self.engine.train_step(images, targets, class_scale=self.tfidf)
            return {"classification": losses[0], "bbox_regression": losses[1]}
        out = self.engine.forward(images, training=False)
        p = self.engine._last_plan
        cls = list(out["cls_logits"].split(p.level_rows, dim=1))
        reg = list(out["bbox_regression"].split(p.level_rows, dim=1))
        det = retinanet_postprocess_detections(cls, reg, p.anchors_per_level, shapes, tfidf_post=self.tfidf_post, score_thresh=self.score_thresh,
                                               topk_candidates=self.topk_candidates, nms_thresh=self.nms_thresh,
                                               detections_per_img=self.detections_per_img)
        if original_image_sizes is not None:
            det = self.transform.postprocess(det, shapes, original_image_sizes)                                           # retinanet.py:567
        return det


def retinanet_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=False, trainable_backbone_layers=None, tfidf=None,
                           **kwargs):
    """retinanet.py:583-660.  No network here: `pretrained*` must be False; load weights with `load_state_dict`."""
    if pretrained or pretrained_backbone:
        raise NotImplementedError("no network access: load a reference state_dict with model.load_state_dict(...)")
    if trainable_backbone_layers is None:
        trainable_backbone_layers = 3           # _validate_trainable_layers default (backbone_utils.py:127-139)
    return RetinaNet(num_classes, trainable_backbone_layers, tfidf=tfidf, **kwargs)


def retinanet_resnet101_fpn(num_classes=1204, trainable_backbone_layers=3, tfidf=None, **kwargs):
    """BASELINE config 5 (RetinaNet ResNet-101-FPN on LVIS, 1203 classes + background): `resnet_fpn_backbone('resnet101', ...)`
    (backbone_utils.py:67-110) under the same RetinaNet head."""
    return RetinaNet(num_classes, trainable_backbone_layers, tfidf=tfidf, body="resnet101", **kwargs)
