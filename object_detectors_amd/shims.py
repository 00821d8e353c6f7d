"""Import-name shims: make the reference's own import statements resolve to the MI355X path, so its scripts run without an edit.

    import object_detectors_amd.shims as shims
    shims.install()                    # before the reference script's own imports

What gets registered (the call sites of the hot path):

  yolo side        nets.yolo_forw.YOLOForw, nets.yolohead.YoloHead, nets.backbone.backbone_fn,
                   utilities.helper.{bbox_iou, nms_majority, get_abs_coord}, procedures.test_one_epoch.{postprocess, to_coco_results},
                   procedures.train_one_epoch.{get_new_scale, multiscale_batch}, procedures.initialize.{save_model, load_checkpoint}
  torchvision side torchvision.ops.boxes.{box_iou, nms, batched_nms, clip_boxes_to_image, remove_small_boxes},
                   torchvision.ops.{sigmoid_focal_loss, roi_align, MultiScaleRoIAlign, boxes},
                   tvision.{retinanet, frcnn, _utils, anchor_utils, rpn, roi_heads, transform, image_list}

Two modes per name.  If the real module is importable (running inside the reference tree, or with torchvision installed) it is imported
and only the hot-path attributes are REPLACED, so everything else it offers (collate functions, label maps, datasets ...) keeps working.
If it is not importable, the mirror module itself is registered under that name.  `uninstall()` restores what was there before."""
import importlib
import sys
import types

_saved = {}          # name -> previous sys.modules entry (or None)
_patched = []        # (module, attribute, previous value or _MISSING)
_MISSING = object()

_YOLO_ATTRS = {
    "utilities.helper": ("object_detectors_amd.yolo.utilities.helper", ("bbox_iou", "nms_majority", "get_abs_coord")),
    "nets.yolo_forw": ("object_detectors_amd.yolo.nets.yolo_forw", ("YOLOForw",)),
    "nets.yolohead": ("object_detectors_amd.yolo.nets.yolohead", ("YoloHead",)),
    "nets.backbone": ("object_detectors_amd.yolo.nets.backbone", ("backbone_fn",)),
    "procedures.test_one_epoch": ("object_detectors_amd.yolo.procedures.test_one_epoch", ("postprocess", "to_coco_results")),
    "procedures.train_one_epoch": ("object_detectors_amd.yolo.procedures.train_one_epoch", ("get_new_scale", "multiscale_batch")),
    "procedures.initialize": ("object_detectors_amd.yolo.procedures.initialize", ("save_model", "load_checkpoint")),
}
_TV_ATTRS = {
    "torchvision.ops.boxes": ("object_detectors_amd.tvision.boxes", ("box_iou", "nms", "batched_nms", "clip_boxes_to_image", "remove_small_boxes")),
    "torchvision.ops": (None, ()),          # filled below: names come from several mirror modules
}
_TVISION = ("retinanet", "frcnn", "_utils", "anchor_utils", "rpn", "roi_heads", "transform", "coco_eval", "boxes", "postprocess", "roi_align", "focal_loss")


def _remember(name):
    if name not in _saved:
        _saved[name] = sys.modules.get(name)


def _patch(mod, attr, value):
    _patched.append((mod, attr, getattr(mod, attr, _MISSING)))
    setattr(mod, attr, value)


def _try_import(name):
    try:
        return importlib.import_module(name)
    except Exception:  # noqa: BLE001  (missing package, or a reference module whose own imports are unavailable)
        return None


def _package(name):
    """an empty package under `name` unless something importable is already there"""
    mod = sys.modules.get(name) or _try_import(name)
    if mod is None:
        _remember(name)
        mod = types.ModuleType(name)
        mod.__path__ = []
        sys.modules[name] = mod
    return mod


def _register(name, mirror_name, attrs):
    mirror = importlib.import_module(mirror_name)
    parent, _, leaf = name.rpartition(".")
    if parent:
        _package(parent)
    real = sys.modules.get(name) or _try_import(name)
    if real is not None and real is not mirror:
        for a in attrs:
            _patch(real, a, getattr(mirror, a))
        mod = real
    else:
        _remember(name)
        sys.modules[name] = mirror
        mod = mirror
    if parent:
        _patch(sys.modules[parent], leaf, mod)
    return mod


def install(yolo=True, torchvision=True):
    """Register / patch the import names listed in the module docstring.  Idempotent."""
    if yolo:
        for name, (mirror, attrs) in _YOLO_ATTRS.items():
            _register(name, mirror, attrs)
    if torchvision:
        _package("torchvision")
        ops_mod = _package("torchvision.ops")
        _patch(sys.modules["torchvision"], "ops", ops_mod)
        boxes = _register("torchvision.ops.boxes", *_TV_ATTRS["torchvision.ops.boxes"])
        from .tvision import focal_loss, roi_align
        for attr, value in (("boxes", boxes), ("sigmoid_focal_loss", focal_loss.sigmoid_focal_loss), ("roi_align", roi_align.roi_align),
                            ("MultiScaleRoIAlign", roi_align.MultiScaleRoIAlign), ("box_iou", boxes.box_iou), ("nms", boxes.nms),
                            ("batched_nms", boxes.batched_nms), ("clip_boxes_to_image", boxes.clip_boxes_to_image),
                            ("remove_small_boxes", boxes.remove_small_boxes)):
            _patch(ops_mod, attr, value)
        # the reference's torchvision_models/tvision package (a namespace package there): the whole mirror package stands in for it
        from . import tvision as mirror_pkg
        _remember("tvision")
        sys.modules["tvision"] = mirror_pkg
        for sub in _TVISION:
            m = importlib.import_module(f"object_detectors_amd.tvision.{sub}")
            _remember(f"tvision.{sub}")
            sys.modules[f"tvision.{sub}"] = m
        from .tvision import transform
        img_list = types.ModuleType("tvision.image_list")
        img_list.ImageList = transform.ImageList
        _remember("tvision.image_list")
        sys.modules["tvision.image_list"] = img_list


def uninstall():
    while _patched:
        mod, attr, prev = _patched.pop()
        if prev is _MISSING:
            try:
                delattr(mod, attr)
            except AttributeError:
                pass
        else:
            setattr(mod, attr, prev)
    for name, prev in list(_saved.items()):
        if prev is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = prev
    _saved.clear()
