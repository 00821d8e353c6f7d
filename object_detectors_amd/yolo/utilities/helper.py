"""Mirror of yolo/utilities/helper.py (reference) for the hot-path functions: same names,
argument meaning and return values; computation in libmi355det.so."""
import torch

from ... import ops


def get_abs_coord(box):
    """helper.py:203-217 — xcycwh -> xyxy on the last axis ([n,4] or [b,n,4])."""
    x1 = box[..., 0] - box[..., 2] / 2
    y1 = box[..., 1] - box[..., 3] / 2
    x2 = box[..., 0] + box[..., 2] / 2
    y2 = box[..., 1] + box[..., 3] / 2
    return torch.stack((x1, y1, x2, y2), axis=-1)


def bbox_iou(bb1, bb2, iou_type, CUDA=True, xcycwh=True):
    """helper.py:221-277.  `CUDA` is kept for signature compatibility (inputs must be on the GPU)."""
    return ops.bbox_iou(bb1, bb2, iou_type, xcycwh)


def nms_majority(P, thresh_iou=0.6, num_classes=None):
    """helper.py:280-382 — greedy NMS with majority-vote relabelling of the kept box.

    P [n,6] = (x1,y1,x2,y2,score,label).  Returns the kept rows [k,6] in keep order.  Like the
    reference, the relabel is written back into P (helper.py:374-375 mutates its input).
    """
    if P.dim() != 2 or P.shape[1] != 6:
        raise ValueError("nms_majority expects [n,6]")
    n = P.shape[0]
    if n == 0:
        raise RuntimeError("stack expects a non-empty TensorList")   # torch.stack([]) in the reference
    if n > 131072:
        raise ValueError("nms_majority: at most 131072 boxes per call (the n*n/8-byte suppression mask is 2 GiB at that size)")
    if num_classes is None:
        num_classes = int(P[:, 5].max().item()) + 1
    count = torch.full((1,), n, device=P.device, dtype=torch.int32)
    rows, idx, kept = ops.nms_majority_batched(P.unsqueeze(0), count, thresh_iou, num_classes)
    k = int(kept.item())
    out = rows[0, :k]
    P[idx[0, :k].long(), 5] = out[:, 5]
    return out
