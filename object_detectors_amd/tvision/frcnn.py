"""Mirror of the reference `FasterRCNN` / `fasterrcnn_resnet50_fpn` (tvision/frcnn.py:21-236,328-400, generalized_rcnn.py) over the
MI355X kernels (BASELINE config 4).

    model = fasterrcnn_resnet50_fpn(num_classes=91)
    losses = model(images, targets)     # {'loss_classifier','loss_box_reg','loss_objectness','loss_rpn_box_reg'}, backward already done
    detections = model(images)          # eval: [{'boxes','labels','scores'}]

Where the work runs:
  * ResNet-FPN body, FPN and the RPN head: `tvision/engine.py:FasterRCNNEngine` (MFMA convolutions, fused FrozenBN/ReLU/identity);
  * anchors, proposal decoding, per-level top-k, clipping, small-box filter, NMS: HIP kernels (`postprocess.rpn_filter_proposals`);
  * RPN / RoI target assignment: fused IoU+Matcher kernels (`tvision/rpn.py`, `tvision/roi_heads.py`); the samplers stay in torch
    (SURVEY 8 row a19);
  * MultiScaleRoIAlign forward/backward: `mi355det_roi_align`;
  * TwoMLPHead / FastRCNNPredictor (frcnn.py:238-290): `tvision/linear.py:MfmaLinear` - the library's MFMA 1x1-convolution kernels
    (forward with bias / ReLU epilogue, data gradient, weight gradient) on [512*N, 12544] bf16 operands; their parameters are ordinary
    fp32 torch parameters with nn.Linear's names, the backbone's live in `model.engine.flat_w` (optimise with
    `optim.FlatSGD.for_engine(model.engine)` + a torch optimizer over `model.head_parameters()`).
Inputs as in tvision/retinanet.py: a list of [3,H,W] images goes through the GPU GeneralizedRCNNTransform (generalized_rcnn.py:78-79,110), a ready
[N,3,H,W] batch skips it.
"""

import torch
from torch import nn

from .. import ops
from ._utils import BoxCoder
from .linear import MfmaLinear
from .engine import IMAGE_MEAN, IMAGE_STD, FasterRCNNEngine
from .postprocess import (roi_heads_postprocess_detections, roi_heads_postprocess_detections_batch, rpn_filter_proposals,
                          rpn_proposals_fused)
from .roi_align import MultiScaleRoIAlign
from .roi_heads import RoIHeadTargets, fastrcnn_loss, minibatch_tfidf
from .rpn import RPNTargets
from .transform import GeneralizedRCNNTransform


# Route switches, module attributes on purpose: the composed routes stay as the bit-exact references of tests/test_gpu_proposals.py and
# tools/bench_frcnn.py (monkeypatch / assignment), not as production knobs (round 3 read them from the environment).
_RPN_FUSED = True           # False: box_decode of every anchor + the torch-composed proposal filter
_ROI_FUSED = True           # False: the per-image torch-composed select_training_samples
_ROI_DET_FUSED = True       # False: inference through proposal lists and the per-image post-processing
_RPN_LOSS_FUSED = True      # False: the autograd-composed RPN losses


class TwoMLPHead(nn.Module):
    """frcnn.py:238-262."""

    def __init__(self, in_channels, representation_size):
        super().__init__()
        self.fc6 = MfmaLinear(in_channels, representation_size, relu=True)          # F.relu(self.fc6(x)) with the ReLU in the GEMM epilogue
        self.fc7 = MfmaLinear(representation_size, representation_size, relu=True)

    def forward(self, x):
        x = x.flatten(start_dim=1)
        return self.fc7(self.fc6(x))


class FastRCNNPredictor(nn.Module):
    """frcnn.py:265-290."""

    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.cls_score = MfmaLinear(in_channels, num_classes)
        self.bbox_pred = MfmaLinear(in_channels, num_classes * 4)

    def forward(self, x):
        if x.dim() == 4:
            assert list(x.shape[2:]) == [1, 1]
        x = x.flatten(start_dim=1)
        return self.cls_score(x), self.bbox_pred(x)


class FasterRCNN(nn.Module):
    def __init__(self, num_classes=91, trainable_backbone_layers=3, tfidf=None,
                 rpn_pre_nms_top_n_train=2000, rpn_pre_nms_top_n_test=1000, rpn_post_nms_top_n_train=2000, rpn_post_nms_top_n_test=1000,
                 rpn_nms_thresh=0.7, rpn_fg_iou_thresh=0.7, rpn_bg_iou_thresh=0.3, rpn_batch_size_per_image=256, rpn_positive_fraction=0.5,
                 rpn_score_thresh=0.0, box_score_thresh=0.05, box_nms_thresh=0.5, box_detections_per_img=100, box_fg_iou_thresh=0.5,
                 box_bg_iou_thresh=0.5, box_batch_size_per_image=512, box_positive_fraction=0.25, bbox_reg_weights=None, loss_type="ce",
                 device=None, seed=0, body="resnet50", min_size=800, max_size=1333, image_mean=None, image_std=None):
        super().__init__()
        self.engine = FasterRCNNEngine(trainable_backbone_layers, device=device, seed=seed, body=body)
        self.transform = GeneralizedRCNNTransform(min_size, max_size, image_mean or list(IMAGE_MEAN), image_std or list(IMAGE_STD))   # frcnn.py:232-236
        dev = self.engine.device
        self.rpn_targets = RPNTargets(rpn_fg_iou_thresh, rpn_bg_iou_thresh, rpn_batch_size_per_image, rpn_positive_fraction)
        self.rpn_pre = dict(training=rpn_pre_nms_top_n_train, testing=rpn_pre_nms_top_n_test)
        self.rpn_post = dict(training=rpn_post_nms_top_n_train, testing=rpn_post_nms_top_n_test)
        self.rpn_nms_thresh, self.rpn_score_thresh = rpn_nms_thresh, rpn_score_thresh
        weights = bbox_reg_weights or (10., 10., 5., 5.)
        self.roi_targets = RoIHeadTargets(box_fg_iou_thresh, box_bg_iou_thresh, box_batch_size_per_image, box_positive_fraction, weights)
        self.box_roi_pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
        self.box_head = TwoMLPHead(256 * 7 * 7, 1024).to(dev)
        self.box_predictor = FastRCNNPredictor(1024, num_classes).to(dev)
        self.box_score_thresh, self.box_nms_thresh, self.box_detections_per_img = box_score_thresh, box_nms_thresh, box_detections_per_img
        self.bbox_reg_weights = weights
        # ---- the tf-idf state of the reference RoIHeads (roi_heads.py:566-572; the dict is built in detection/train.py:103-135):
        #      'values' [1,K] row multiplied into the class logits (training: self.tfidf, replaced per batch when 'mini_batch'; inference:
        #      the untouched copy tfidf_post), 'classification_weights' [K] or None (cross-entropy class weights), 'loss_function'
        if tfidf is None:
            tfidf = {}
        unknown = set(tfidf) - {"values", "num_classes", "mini_batch", "tfidf_norm", "loss_function", "classification_weights"}
        if unknown:
            raise ValueError(f"FasterRCNN: unsupported tfidf fields {sorted(unknown)}")
        if tfidf.get("num_classes", num_classes) != num_classes:
            raise ValueError("FasterRCNN: tfidf['num_classes'] differs from num_classes")
        values = tfidf.get("values")
        values = torch.ones(1, num_classes) if values is None else values
        self.tfidf = values.detach().to(dev).float().reshape(1, num_classes)
        self.tfidf_post = self.tfidf.clone()
        self.tfidf_mini_batch = bool(tfidf.get("mini_batch", False))
        self.tfidf_norm = tfidf.get("tfidf_norm", 0)
        self.loss_function_name = tfidf.get("loss_function", loss_type)
        if self.loss_function_name not in ops.FRCNN_LOSS_TYPES:
            raise ValueError(f"FasterRCNN: unknown loss function {self.loss_function_name!r} (reference: {sorted(ops.FRCNN_LOSS_TYPES)})")
        cw = tfidf.get("classification_weights")
        self.classification_weights = None if cw is None else cw.detach().to(dev).float()
        self.loss_type = self.loss_function_name
        self.num_classes = num_classes
        self.rpn_coder = BoxCoder((1.0, 1.0, 1.0, 1.0))

    def head_parameters(self):
        return list(self.box_head.parameters()) + list(self.box_predictor.parameters())

    def state_dict(self, *a, **k):
        sd = self.engine.reference_state_dict()
        for k2, v in self.box_head.state_dict().items():
            sd["roi_heads.box_head." + k2] = v
        for k2, v in self.box_predictor.state_dict().items():
            sd["roi_heads.box_predictor." + k2] = v
        return sd

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.engine.load_reference_state_dict(sd)
        self.box_head.load_state_dict({k[len("roi_heads.box_head."):]: v for k, v in sd.items() if k.startswith("roi_heads.box_head.")})
        self.box_predictor.load_state_dict({k[len("roi_heads.box_predictor."):]: v for k, v in sd.items()
                                            if k.startswith("roi_heads.box_predictor.")})

    # ------------------------------------------------------------------
    def _proposals(self, out, plan, image_shapes, counts_out=None):
        """rpn.py:336-351: decode every anchor with the (detached) deltas, then filter_proposals.  With `counts_out` (int32 [N] on the
        device) the padded form [N, post, 4] comes back and nothing is read by the host."""
        n = out["cls_logits"].shape[0]
        mode = "training" if self.training else "testing"
        if _RPN_FUSED:
            return rpn_proposals_fused(out["bbox_regression"], out["cls_logits"], plan.anchors, image_shapes, plan.level_rows, self.rpn_pre[mode],
                                       self.rpn_post[mode], self.rpn_nms_thresh, self.rpn_score_thresh,
                                       xform_clip=self.rpn_coder.bbox_xform_clip, counts_out=counts_out)
        deltas = out["bbox_regression"].detach().reshape(-1, 4)
        anchors = plan.anchors.repeat(n, 1)
        proposals = ops.box_decode(deltas, anchors, (1.0, 1.0, 1.0, 1.0), self.rpn_coder.bbox_xform_clip).reshape(n, -1, 4)
        return rpn_filter_proposals(proposals, out["cls_logits"].detach().reshape(n, -1), image_shapes, plan.level_rows, self.rpn_pre[mode],
                                    self.rpn_post[mode], self.rpn_nms_thresh, self.rpn_score_thresh)

    def _detect_padded(self, out, plan, image_shapes):
        """Inference with the proposals kept padded on the device from the RPN to the detections: proposal filter, RoIAlign, box head and
        `postprocess_detections` without a host read in between - ONE read at the end (the list route reads the proposal counts, then three
        counts per image in the post-processing).  None when an image has too many candidates for the one-call post-processing."""
        n = out["cls_logits"].shape[0]
        dev = out["cls_logits"].device
        counts = torch.empty(n, device=dev, dtype=torch.int32)
        with torch.no_grad():
            boxes, _scores = self._proposals(out, plan, image_shapes, counts_out=counts)
            p = boxes.shape[1]
            key = (n, p, str(dev))
            if getattr(self, "_roi_ids_key", None) != key:
                self._roi_ids = torch.arange(n, device=dev, dtype=torch.float32).repeat_interleave(p)[:, None]
                self._roi_ids_key = key
            rois = torch.cat([self._roi_ids, boxes.reshape(-1, 4)], 1)
            x = self.box_roi_pool.forward_nhwc(self.engine.feature_maps_nhwc(4), rois, image_shapes)
            cls, reg = self.box_predictor(self.box_head(x))
            res = roi_heads_postprocess_detections_batch(cls, reg, boxes, counts, image_shapes, self.tfidf_post, self.box_score_thresh,
                                                         self.box_nms_thresh, self.box_detections_per_img, self.bbox_reg_weights, self.loss_type)
        if res is None:
            return None
        return [{"boxes": bb, "labels": ll, "scores": ss} for bb, ll, ss in zip(*(res[0], res[2], res[1]))]

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")          # generalized_rcnn.py:60-61
        original_image_sizes = None
        if isinstance(images, (list, tuple)):
            original_image_sizes = [(int(i.shape[-2]), int(i.shape[-1])) for i in images]      # generalized_rcnn.py:72-76
            self.transform.train(self.training)
            image_list, targets = self.transform(images, targets)                              # generalized_rcnn.py:78
            images, image_shapes = image_list.tensors, list(image_list.image_sizes)
            self.engine.normalize = False              # normalised before padding by the transform: padded pixels are exact zeros
        else:
            image_shapes = [(images.shape[-2], images.shape[-1])] * images.shape[0]
            self.engine.normalize = True
        if self.training:
            for t in targets:
                b = t["boxes"]
                if b.dim() != 2 or b.shape[-1] != 4:
                    raise ValueError("Expected target boxes to be a tensor of shape [N, 4], got {:}.".format(b.shape))
                if b.numel() and bool((b[:, 2:] <= b[:, :2]).any()):
                    raise ValueError("All bounding boxes should have positive height and width.")
        n = images.shape[0]
        if targets is not None and self.tfidf_mini_batch:         # roi_heads.py:801-809
            self.tfidf = minibatch_tfidf(targets, self.num_classes, self.tfidf_norm).to(images.device).float().reshape(1, -1)
        rpn_side = None
        if self.training:
            # RPN target assignment + sampling (rpn.py:179-213,296-300) need only the anchors and the ground truth, and they are host-bound
            # (the RNG sampler reads counts back): issue them on a side stream so that they run under the network forward instead of after it
            plan0 = self.engine.plan(images.shape[0], images.shape[2], images.shape[3], True)
            cur = torch.cuda.current_stream(images.device)
            if not hasattr(self, "_tgt_stream"):
                self._tgt_stream = torch.cuda.Stream(device=images.device)
            self._tgt_stream.wait_stream(cur)
        out = self.engine.forward(images, training=self.training)
        plan = self.engine._last_plan
        rpn_losses = None
        if self.training:
            assert plan is plan0
            fwd_done = torch.cuda.Event()
            fwd_done.record(cur)
            with torch.cuda.stream(self._tgt_stream):
                rpn_side = self.rpn_targets.prepare([plan.anchors] * n, targets)
                # (measured and removed: the RPN losses on this stream too, beside the proposal kernels - no gain, profiles/r03_ab_results.md)
        if not self.training and _RPN_FUSED and _ROI_DET_FUSED:
            det = self._detect_padded(out, plan, image_shapes)
            if det is not None:
                if original_image_sizes is not None:
                    det = self.transform.postprocess(det, image_shapes, original_image_sizes)      # generalized_rcnn.py:110
                return det
        fused = self.training and _RPN_FUSED and _ROI_FUSED and self.roi_targets.fused_ok(n, self.rpn_post["training"], targets)
        if fused:     # proposals stay padded on the device; their counts are read together with the sampler's counts (one host read in all)
            meta = torch.empty(3 * n, device=images.device, dtype=torch.int32)
            boxes, _scores = self._proposals(out, plan, image_shapes, counts_out=meta[:n])
        else:
            boxes, _scores = self._proposals(out, plan, image_shapes)
        feats = self.engine.feature_maps_nhwc(4)       # the engine's bf16 NHWC buffers themselves
        if not self.training:
            with torch.no_grad():
                x = self.box_roi_pool.forward_nhwc(feats, boxes, image_shapes)
                cls, reg = self.box_predictor(self.box_head(x))
                b, s, l = roi_heads_postprocess_detections(cls, reg, boxes, image_shapes, self.tfidf_post, self.box_score_thresh,
                                                           self.box_nms_thresh, self.box_detections_per_img, self.bbox_reg_weights, self.loss_type)
            det = [{"boxes": bb, "labels": ll, "scores": ss} for bb, ll, ss in zip(b, l, s)]
            if original_image_sizes is not None:
                det = self.transform.postprocess(det, image_shapes, original_image_sizes)      # generalized_rcnn.py:110
            return det
        # ---- RoI heads (roi_heads.py:783-848): sample, pool, two FC layers, predictor, Fast R-CNN loss
        if fused:
            proposals, _mi, labels, reg_targets, _per_image = self.roi_targets.select_training_samples_fused(boxes, meta, targets)
            labels, reg_targets = [labels], [reg_targets]
        else:
            proposals, _mi, labels, reg_targets = self.roi_targets.select_training_samples([b.detach() for b in boxes], targets)
        x = self.box_roi_pool.forward_nhwc(feats, proposals, image_shapes)
        cls, reg = self.box_predictor(self.box_head(x))
        # roi_heads.py:826-827: fastrcnn_loss(self.tfidf * class_logits, ..., weights=self.classification_weights, loss_type=...)
        loss_cls, loss_box = fastrcnn_loss(cls, reg, labels, reg_targets, weights=self.classification_weights, loss_type=self.loss_function_name,
                                           class_scale=self.tfidf)
        # ---- RPN losses on leaf copies of the engine's outputs (their .grad is the engine's head gradient).  Issued AFTER the RoI branch:
        # they do not feed it, and their host time then hides behind the RoI kernels instead of sitting in front of them with the device idle
        torch.cuda.current_stream(images.device).wait_stream(self._tgt_stream)
        obj_grad = dl_grad = None
        if rpn_losses is None and _RPN_LOSS_FUSED and rpn_side["sampled"].numel():
            # compute_loss and its gradients in ONE launch, outside autograd (the autograd form: ~25 launches, two of them sort-based
            # index_put(accumulate) for the backward of the gathers); the loss weights are 1, as in the sum below
            rpn_losses, obj_grad, dl_grad = self.rpn_targets.losses_prepared_fused(out["cls_logits"].reshape(-1, 1), out["bbox_regression"].reshape(-1, 4),
                                                                                   rpn_side)
        elif rpn_losses is None:
            obj = out["cls_logits"].detach().reshape(-1, 1).requires_grad_(True)
            dl = out["bbox_regression"].detach().reshape(-1, 4).requires_grad_(True)
            rpn_losses = self.rpn_targets.losses_prepared(obj, dl, rpn_side)
        losses = {"loss_classifier": loss_cls, "loss_box_reg": loss_box}
        losses.update(rpn_losses)
        if obj_grad is not None:
            (loss_cls + loss_box).backward()
        else:
            sum(losses.values()).backward()
            obj_grad, dl_grad = obj.grad, dl.grad
        if getattr(self, "head_grad_sync", None) is not None:     # data parallel: parallel.ParamGradSync over head_parameters(), overlapped with
            self.head_grad_sync.reduce()                          # the whole backbone backward below
        self.engine.backward(obj_grad, dl_grad, [f.grad for f in feats])
        return {k: v.detach() for k, v in losses.items()}


def fasterrcnn_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=False, trainable_backbone_layers=None, tfidf=None,
                            **kwargs):
    """frcnn.py:328-400.  No network here: `pretrained*` must be False; load weights with `load_state_dict`."""
    if pretrained or pretrained_backbone:
        raise NotImplementedError("no network access: load a reference state_dict with model.load_state_dict(...)")
    if trainable_backbone_layers is None:
        trainable_backbone_layers = 3
    return FasterRCNN(num_classes, trainable_backbone_layers, tfidf=tfidf, **kwargs)
