"""Shared builders that regenerate the golden fixtures' inputs from their detrand seeds."""
import numpy as np

from oracle import detrand
from oracle.yolo_oracle import YoloSpec

YOLO_CASES = ["coco128", "coco128_idf", "coco128_iou", "coco128_diou", "coco128_ciou", "lvis96_a6",
              "coco416", "coco640", "coco128_cw", "coco128_batchidf", "coco128_bce", "coco128_eql", "coco128_mean", "coco128_bce_mean"]
YOLO_CLASS_LOSS = {"coco128_bce": 0, "coco128_eql": 2, "coco128_bce_mean": 0}          # class_loss of the case (default 1 = CrossEntropy)
YOLO_FULL = YOLO_CASES[:6] + YOLO_CASES[8:]


def synth_targets(seed, ms, num_classes):
    out = []
    for b, m in enumerate(ms):
        xy = detrand.uniform(seed + 17 * b, (m, 2), 0.2, 0.8)
        wh = detrand.uniform(seed + 17 * b + 5, (m, 2), 0.02, 0.32)
        lab = detrand.randint(seed + 17 * b + 9, (m,), 0, num_classes)
        out.append((np.concatenate([xy, wh], 1).astype(np.float32), lab))
    return out


def synth_heads(seed, bs, na, nc, grids):
    return [detrand.uniform(seed + k, (bs, na * (5 + nc), g, g), -3.0, 3.0) for k, g in enumerate(grids)]


def yolo_case(g3, name):
    """-> (spec, heads, targets) exactly as tools/make_golden.py built them."""
    seed, C, img, iou_type, na, bs = [int(v) for v in g3[name + "_meta"]]
    grids = [int(v) for v in g3[name + "_grids"]]
    ms = [int(v) for v in g3[name + "_ms"]]
    anchors = g3[name + "_anchors"].tolist()
    idf = g3[name + "_idf"] if (name + "_idf") in g3.files else None
    if (name + "_batch_idf") in g3.files:        # tfidf_batch: the row the reference computed from this batch (IDFTransformer.forward, p-normalised)
        idf = g3[name + "_batch_idf"]
    cw = g3[name + "_cw"] if (name + "_cw") in g3.files else None
    heads = synth_heads(seed, bs, na, C, grids)
    targets = synth_targets(seed + 50, ms, C)
    if name == "coco128":
        targets[0][0][1] = targets[0][0][0] + np.float32(1e-3)
    eq_mask = g3[name + "_eq_mask"] if (name + "_eq_mask") in g3.files else None
    spec = YoloSpec(anchors, C, img, iou_type=iou_type, idf_logits=idf, class_weights=cw, class_loss=YOLO_CLASS_LOSS.get(name, 1),
                    reduction="mean" if name.endswith("mean") else "sum", eq_mask=eq_mask)
    spec.img_freq = g3[name + "_img_freq"] if (name + "_img_freq") in g3.files else None     # what custom.EQLoss derives eq_mask from
    return spec, heads, targets
