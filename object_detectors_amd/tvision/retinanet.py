"""Mirror of the reference `RetinaNet` / `retinanet_resnet50_fpn` (tvision/retinanet.py:249-660) over the MI355X engine.

    model = retinanet_resnet50_fpn(num_classes=91, tfidf={...})      # same call as the reference's detection/train.py
    losses = model(images, targets)        # training: {'classification': ..., 'bbox_regression': ...} (backward already done)
    detections = model(images)             # eval: [{'boxes','scores','labels'}]

Differences (documented in INTEGRATION.md): images must already be resized / batched to one [N,3,H,W] tensor with H, W
multiples of 32 (GeneralizedRCNNTransform's resize is data-pipeline work outside the hot path; its normalisation is fused
into the stem); in training mode the call runs the fused forward + loss + backward and leaves the gradients in
`model.engine.flat_g` (use `object_detectors_amd.optim.FlatSGD.for_engine(model.engine)`), returning detached losses.
"""
import torch
from torch import nn

from .engine import RetinaNetEngine
from .postprocess import retinanet_postprocess_detections


class RetinaNet(nn.Module):
    def __init__(self, num_classes=91, trainable_backbone_layers=3, score_thresh=0.05, nms_thresh=0.5, detections_per_img=300,
                 topk_candidates=1000, tfidf=None, device=None, seed=0, body="resnet50"):
        super().__init__()
        self.engine = RetinaNetEngine(num_classes, 9, trainable_backbone_layers, device=device, seed=seed, body=body)
        self.score_thresh, self.nms_thresh = score_thresh, nms_thresh
        self.detections_per_img, self.topk_candidates = detections_per_img, topk_candidates
        self.tfidf = None if tfidf is None else tfidf["values"].to(self.engine.device).float()
        self.tfidf_post = self.tfidf

    def state_dict(self, *a, **k):
        return self.engine.reference_state_dict()

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.engine.load_reference_state_dict(sd)

    @staticmethod
    def _batch(images):
        if isinstance(images, (list, tuple)):
            if len({tuple(i.shape) for i in images}) != 1:
                raise NotImplementedError("images of different sizes: resize/pad them to one size first (transform.py:88-118 is not on the GPU path)")
            images = torch.stack(list(images))
        return images

    def forward(self, images, targets=None):
        images = self._batch(images)
        if self.training:
            if targets is None:
                raise ValueError("In training mode, targets should be passed")          # retinanet.py:489-490
            for t in targets:
                b = t["boxes"]
                if b.dim() != 2 or b.shape[-1] != 4:
                    raise ValueError("Expected target boxes to be a tensor of shape [N, 4], got {:}.".format(b.shape))   # retinanet.py:495-501
                if b.numel() and bool(((b[:, 2:] <= b[:, :2]).any())):
                    raise ValueError("All bounding boxes should have positive height and width.")                        # retinanet.py:515-525
            losses = self.engine.train_step(images, targets, class_scale=self.tfidf)
            return {"classification": losses[0], "bbox_regression": losses[1]}
        out = self.engine.forward(images, training=False)
        p = self.engine._last_plan
        cls = list(out["cls_logits"].split(p.level_rows, dim=1))
        reg = list(out["bbox_regression"].split(p.level_rows, dim=1))
        shapes = [(images.shape[-2], images.shape[-1])] * images.shape[0]
        return retinanet_postprocess_detections(cls, reg, p.anchors_per_level, shapes, tfidf_post=self.tfidf_post, score_thresh=self.score_thresh,
                                                topk_candidates=self.topk_candidates, nms_thresh=self.nms_thresh,
                                                detections_per_img=self.detections_per_img)


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
