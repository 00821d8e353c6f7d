"""Mirror of the post-processing (:24-37) and of the result formatting (:41-66) inside yolo/procedures/test_one_epoch.py."""
import itertools

import torch

from ... import ops
from ..._lib import check, lib, ptr, stream_ptr


def postprocess(predictions, confidence=0.1, iou_threshold=0.6, num_classes=None, criterion=None):
    """predictions: decoded [bs,N,5+C] (YOLOForw inference output).  Returns the list of per-image
    [k,6] tensors (x1,y1,x2,y2,score,label) for images with at least one box above `confidence`, exactly
    like `pred_final` in the reference (which calls nms_majority with its default 0.6 threshold)."""
    bs, n, attrs = predictions.shape
    num_classes = num_classes or (attrs - 5)
    score = label = None
    fused = getattr(criterion, "last_decode_scores", None) if criterion is not None else None
    if fused is not None and fused[0]() is predictions and fused[1] == predictions._version:
        score, label = fused[2], fused[3]       # computed by the decode kernel for THIS tensor, unmodified since: the 85-wide rows are not re-read
        criterion.last_decode_scores = None     # consumed
    cand, count = ops.yolo_candidates(predictions, confidence, score=score, label=label)
    counts = count.tolist()                     # one host sync for the ragged python-list output
    if max(counts) > cand.shape[1]:
        raise RuntimeError(f"more than {cand.shape[1]} boxes above the confidence threshold in one image")
    max_n = max(1, max(counts))
    cand = cand[:, :max_n].contiguous()
    rows, _idx, kept = ops.nms_majority_batched(cand, count, iou_threshold, num_classes)
    kept = kept.tolist()
    return [rows[b, :kept[b]] for b in range(bs) if counts[b] > 0]


def to_coco_results(pred_final, targets, inp_dim, dset_name="coco"):
    """test_one_epoch.py:41-66: the kept detections of a batch -> the list of COCO result dicts
    {'bbox': [x, y, w, h] in the original image's pixels, 'area', 'category_id' (80 -> 91 map for coco, label + 1 otherwise), 'score',
    'image_id'}.  As in the reference, entry i of `pred_final` is paired with `targets[i]` - images without any detection were dropped
    from `pred_final` upstream (:31,37), so after such an image the pairing is shifted; callers that want the true pairing pass the
    targets of the surviving images."""
    results = []
    L = lib()
    for i, atrbs in enumerate(pred_final):
        k = int(atrbs.shape[0])
        if k == 0:
            continue
        rows = atrbs.float().contiguous()
        size = targets[i]["img_size"]
        bbox = torch.empty((k, 4), dtype=torch.float32, device=rows.device)
        area = torch.empty(k, dtype=torch.float32, device=rows.device)
        cat = torch.empty(k, dtype=torch.int64, device=rows.device)
        check(L.mi355det_coco_rows(ptr(rows), 6, ptr(rows[:, 5:]), None, 6, k, float(inp_dim), float(size[0]), float(size[1]), 1,
                                   1 if dset_name == "coco" else 0, ptr(bbox), ptr(area), ptr(cat), stream_ptr()), "coco_rows")
        image_id = targets[i]["image_id"].item()
        temp = [{"bbox": b, "area": a, "category_id": l, "score": s, "image_id": image_id}
                for b, a, l, s in zip(bbox.tolist(), area.tolist(), cat.tolist(), rows[:, 4].tolist())]     # one host copy per image
        results = list(itertools.chain(results, temp))
    return results
