"""object_detectors_amd.shims: the reference's import statements resolve to the mirrors (no GPU needed: nothing is computed)."""
import importlib
import sys
import types

import pytest


@pytest.fixture()
def clean_modules():
    names = [n for n in list(sys.modules) if n.split(".")[0] in ("nets", "utilities", "procedures", "torchvision", "tvision")]
    saved = {n: sys.modules.pop(n) for n in names}
    yield
    from object_detectors_amd import shims
    shims.uninstall()
    for n in [n for n in list(sys.modules) if n.split(".")[0] in ("nets", "utilities", "procedures", "torchvision", "tvision")]:
        sys.modules.pop(n, None)
    sys.modules.update(saved)


def test_reference_import_statements_resolve_to_the_mirrors(clean_modules):
    from object_detectors_amd import shims
    shims.install()
    # yolo/main.py, yolo/procedures/*.py style imports
    from nets.yolo_forw import YOLOForw
    from nets.yolohead import YoloHead
    from utilities import helper
    from nets.backbone import backbone_fn
    from procedures.test_one_epoch import postprocess, to_coco_results
    import object_detectors_amd.yolo.nets.yolo_forw as mf
    import object_detectors_amd.yolo.utilities.helper as mh
    assert YOLOForw is mf.YOLOForw and helper.nms_majority is mh.nms_majority and helper.bbox_iou is mh.bbox_iou
    assert set(backbone_fn) == {"darknet_21", "darknet_53"} and callable(postprocess) and callable(to_coco_results) and YoloHead is not None
    # torchvision_models/tvision/*.py style imports
    from torchvision.ops import boxes as box_ops
    from torchvision.ops import MultiScaleRoIAlign, roi_align, sigmoid_focal_loss
    import torchvision
    import object_detectors_amd.tvision.boxes as mb
    assert box_ops.batched_nms is mb.batched_nms and box_ops.box_iou is mb.box_iou and torchvision.ops.boxes is box_ops
    assert callable(sigmoid_focal_loss) and callable(roi_align) and MultiScaleRoIAlign.__module__.startswith("object_detectors_amd")
    # detection/train.py style imports of the model constructors
    from tvision.retinanet import retinanet_resnet50_fpn
    from tvision.frcnn import fasterrcnn_resnet50_fpn
    from tvision._utils import BoxCoder, Matcher
    from tvision.anchor_utils import AnchorGenerator
    from tvision.image_list import ImageList
    from tvision.transform import GeneralizedRCNNTransform, resize_boxes
    assert all(callable(f) for f in (retinanet_resnet50_fpn, fasterrcnn_resnet50_fpn, BoxCoder, Matcher, AnchorGenerator, ImageList,
                                     GeneralizedRCNNTransform, resize_boxes))
    shims.uninstall()
    assert "nets.yolo_forw" not in sys.modules and "tvision.retinanet" not in sys.modules
    with pytest.raises(ImportError):
        importlib.import_module("nets.yolo_forw")


def test_existing_module_is_patched_not_replaced(clean_modules):
    """When the real module exists (the reference tree on sys.path, or torchvision installed) only the hot-path attributes are swapped."""
    pkg = types.ModuleType("utilities")
    pkg.__path__ = []
    real = types.ModuleType("utilities.helper")
    real.collate_fn = lambda b: b                     # something the mirror does not provide
    real.nms_majority = lambda P, thresh_iou=0.6: "reference"
    sys.modules["utilities"], sys.modules["utilities.helper"] = pkg, real
    from object_detectors_amd import shims
    shims.install(torchvision=False)
    from utilities import helper
    import object_detectors_amd.yolo.utilities.helper as mh
    assert helper is real and helper.nms_majority is mh.nms_majority and helper.collate_fn([1]) == [1]
    shims.uninstall()
    assert real.nms_majority(None) == "reference"
