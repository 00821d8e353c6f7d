"""Mirror of the post-processing inside yolo/procedures/test_one_epoch.py:24-37."""
import torch

from ... import ops


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
