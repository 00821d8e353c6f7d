"""Mirror of yolo/nets/yolo_forw.py: YOLOForw criterion (train loss + inference decode).

Same call signature and return values as the reference (`YOLOForw(out, targets=None)` ->
`(loss, sub_losses[6], stats[5])` or decoded `[bs, N, 5+C]`), but target assignment, the six loss
terms, their gradients and the decode are fused HIP kernels that read the head tensors in place.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from ... import _lib, ops


def _get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


class _YoloLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, targets, *heads):
        out12, grads = mod._loss_impl(heads, targets, want_grad=any(h.requires_grad for h in heads))
        ctx.grads = grads
        ctx.nheads = len(heads)
        ctx.mark_non_differentiable(out12)
        return out12[0].clone(), out12

    @staticmethod
    def backward(ctx, gloss, _g12):
        grads = ctx.grads
        if grads is None:
            return (None, None) + (None,) * ctx.nheads
        return (None, None) + tuple(g * gloss for g in grads)


class YOLOForw(nn.Module):
    def __init__(self, config=None, *, anchors=None, num_classes=None, img_size=None, idf_logits=None, class_weights=None, idf=None,
                 img_freq=None, **kw):
        """Either pass the reference's hydra-style `config` (config.yolo.*, config.dataset.anchors) or
        keyword arguments.  Defaults follow hydra/yolo/head.yaml:4-17."""
        super().__init__()
        ycfg = _get(config, "yolo", {}) if config is not None else {}
        dcfg = _get(config, "dataset", {}) if config is not None else {}

        def opt(name, default):
            return kw.get(name, _get(ycfg, name, default))
        self.anchors = anchors if anchors is not None else _get(dcfg, "anchors")
        self.num_anchors = len(self.anchors)
        self.num_classes = num_classes if num_classes is not None else _get(ycfg, "classes")
        self.bbox_attrs = 5 + self.num_classes
        self.img_size = img_size if img_size is not None else _get(ycfg, "img_size")
        self.ignore_threshold = opt("ignore_threshold", 0.5)
        self.lambda_iou = opt("lambda_iou", 1)
        self.lambda_xy = opt("lambda_xy", 2.5)
        self.lambda_wh = opt("lambda_wh", 2.5)
        self.lambda_conf = opt("lambda_conf", 1.0)
        self.lambda_no_conf = opt("lambda_no_conf", 0.1)
        self.lambda_cls = opt("lambda_cls", 1.0)
        self.reduction = opt("reduction", "sum")
        self.iou_type = opt("iou_type", 1)
        self.alpha, self.gamma = opt("alpha", 0.5), opt("gamma", 1)
        self.class_loss = opt("class_loss", 1)
        # class_loss (yolo_forw.py:69-77): 0 BCEWithLogitsLoss(pos_weight), 1 CrossEntropyLoss(weight) [head.yaml default], 2 custom.EQLoss over
        # the BCE form; reduction 'sum' [default] or 'mean' (any other string behaves like 'mean' in the reference's `== "sum"` tests only for
        # the iou / no-object terms and is rejected by torch's loss constructors, so it is rejected here too)
        if self.class_loss not in (0, 1, 2):
            raise ValueError(f"YOLOForw: class_loss must be 0 (bce), 1 (ce) or 2 (eql), got {self.class_loss!r}")
        if self.reduction not in ("sum", "mean"):
            raise ValueError(f"YOLOForw: {self.reduction!r} is not a valid value for reduction")
        self.device = torch.device("cuda")
        # ---- class re-weighting (yolo_forw.py:33-67): `tfidf` = [weights switch, logits switch]; weights switch 1 = CrossEntropyLoss class
        #      weights from the idf table, 2 = effective-number weights (beta 0.9999 on instance_freq); logits switch 1 = the idf row
        #      multiplies the class logits in loss and decode; `tfidf_norm` p-normalises either; `tfidf_batch` recomputes the logits row
        #      from every training batch (IDFTransformer.forward)
        tf = opt("tfidf", [0, 0])
        variant = opt("tfidf_variant", "smooth")
        self.tfidf_norm = opt("tfidf_norm", 0)
        self.tfidf_batch = bool(opt("tfidf_batch", False))
        self.idf = idf
        if self.idf is None and (tf[0] or tf[1]):
            from ..utilities.custom import IDFTransformer
            self.idf = IDFTransformer(_get(dcfg, "train_annotations"), _get(dcfg, "dset_name", "coco"), device="cpu")
        if self.idf is None and self.tfidf_batch:
            from ..utilities.custom import IDFTransformer
            self.idf = IDFTransformer(num_classes=self.num_classes, device="cpu")
        if class_weights is None and tf[0] == 1:
            class_weights = self.idf.idf_weights[variant].float()
            if self.tfidf_norm != 0:
                class_weights = class_weights / torch.norm(class_weights, p=self.tfidf_norm)
        elif class_weights is None and tf[0] == 2:
            import numpy as np
            freq = self.idf.idf_weights["instance_freq"].cpu().numpy()
            w = (1.0 - 0.9999) / (1.0 - np.power(0.9999, freq))
            class_weights = torch.from_numpy((w / np.sum(w) * len(freq)).astype("float32"))
        if idf_logits is None and tf[1] == 1:
            idf_logits = self.idf.idf_weights[variant].float()
            if self.tfidf_norm != 0:
                idf_logits = idf_logits / torch.norm(idf_logits, p=self.tfidf_norm)
        if class_weights is None:
            self.class_weights = None
        else:
            self.register_buffer("class_weights", torch.as_tensor(class_weights, dtype=torch.float32))
        # custom.EQLoss.eq_mask (custom.py:79-80): classes whose share of the image frequencies is below 0.0045
        if self.class_loss != 2:
            self.eq_mask = None
        else:
            if img_freq is None:
                if self.idf is None:
                    from ..utilities.custom import IDFTransformer
                    self.idf = IDFTransformer(_get(dcfg, "train_annotations"), _get(dcfg, "dset_name", "coco"), device="cpu")
                img_freq = self.idf.idf_weights["img_freq"]
            img_freq = torch.as_tensor(img_freq, dtype=torch.float32)
            self.register_buffer("eq_mask", ((img_freq / img_freq.sum()) < 0.0045).float())
        # idf_logits (yolo_forw.py:38,63-67): scalar 1 or a [C] vector multiplying the class logits
        if idf_logits is None:
            self.idf_logits = None
        else:
            self.register_buffer("idf_logits", torch.as_tensor(idf_logits, dtype=torch.float32))
        self._geom_cache = {}

    def set_img_size(self, img_size):
        self.img_size = img_size

    # ------------------------------------------------------------------------------------
    def _geom(self, grids):
        key = (tuple(grids), self.img_size)
        if key not in self._geom_cache:
            self._geom_cache[key] = ops.make_geom(self.anchors, self.num_classes, self.img_size, grids,
                                                  self.iou_type, self.ignore_threshold)
        return self._geom_cache[key]

    def _idf(self, device):
        if self.idf_logits is None:
            return None
        return self.idf_logits.to(device=device, dtype=torch.float32).contiguous()

    def _loss_impl(self, heads, targets, want_grad, grad_views=None, grad_is_bf16=False, grad_scale=1.0):
        for h in heads:
            if h.shape[2] != h.shape[3]:
                raise ValueError("YOLOForw: square feature maps only (as the reference's grid construction assumes)")
        grids = [int(h.shape[2]) for h in heads]
        geom = self._geom(grids)
        attrs_total = len(self.anchors[0]) * self.bbox_attrs
        hv, keep = ops.head_views(heads, attrs_total)
        dev = keep[0].device
        bs = keep[0].shape[0]
        boxes, labels, off, counts = ops.flatten_targets(targets, dev)
        G = boxes.shape[0]
        if G == 0:
            raise ValueError("YOLOForw: batch without any ground-truth box")
        if self.tfidf_batch:                                  # yolo_forw.py:87-91: the logits row of THIS batch (kept for later decodes, as there)
            row = self.idf(targets).float()
            if self.tfidf_norm != 0:
                row = row / torch.norm(row, p=self.tfidf_norm)
            self.idf_logits = row.to(dev)
        obj_idx, tgt, noobj = ops.yolo_assign(geom, boxes, off, bs, counts)
        grads = None
        gv = None
        if grad_views is not None:
            gv = grad_views
        elif want_grad:
            grads = [torch.zeros_like(t) for t in keep]
            gv, _ = ops.head_views(grads, attrs_total)
        cw = None if self.class_weights is None else self.class_weights.to(device=dev, dtype=torch.float32).contiguous()
        eq = None if self.eq_mask is None else self.eq_mask.to(device=dev, dtype=torch.float32).contiguous()
        cfg = _lib.YoloLossCfg(self.lambda_iou, self.lambda_xy, self.lambda_wh, self.lambda_conf, self.lambda_no_conf,
                               self.lambda_cls, self.alpha, self.gamma, grad_scale, int(grad_is_bf16), None if cw is None else cw.data_ptr(),
                               int(self.class_loss), int(self.reduction == "mean"), None if eq is None else eq.data_ptr())
        self._cw_keep = (cw, eq)
        out12 = ops.yolo_loss(geom, cfg, hv, gv, off, labels, obj_idx, tgt, noobj, self._idf(dev), bs, G)
        self.last_assignment = (obj_idx, tgt, noobj, counts)
        return out12, grads

    def forward(self, input, targets=None):
        heads = list(input)
        if targets is not None:
            loss, out12 = _YoloLossFn.apply(self, targets, *heads)
            return loss, out12[1:7], out12[7:12]
        grids = [int(h.shape[2]) for h in heads]
        geom = self._geom(grids)
        hv, keep = ops.head_views(heads, len(self.anchors[0]) * self.bbox_attrs)
        out, score, label = ops.yolo_decode(geom, hv, self._idf(keep[0].device), keep[0].shape[0], softmax_cls=self.class_loss == 1, want_scores=True)
        # conf*max(cls) / arg-max from the same pass (channels-last heads), picked up by procedures.test_one_epoch.postprocess
        # keyed on the tensor OBJECT (weak reference) and its version counter, not on its address: the caching allocator hands a freed block's
        # address to the next tensor of the same size, which would otherwise match a stale entry
        self.last_decode_scores = (weakref.ref(out), out._version, score, label) if score is not None else None
        return out

    def get_target(self, targets, grids, device=None):
        """YOLOForw.get_target (yolo_forw.py:178-208) for a whole batch: (tgt, obj_mask list, noobj_mask)."""
        dev = device or torch.device("cuda")
        geom = self._geom(list(grids))
        boxes, labels, off, counts = ops.flatten_targets(targets, dev)
        obj_idx, tgt, noobj = ops.yolo_assign(geom, boxes, off, len(targets), counts)
        return tgt, list(torch.split(obj_idx, counts)), noobj.bool()
