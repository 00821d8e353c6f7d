"""Mirror of the torchvision.ops.boxes functions the reference calls:
box_iou, nms, batched_nms, clip_boxes_to_image, remove_small_boxes — HIP kernels underneath."""
import torch

from .. import ops


def box_iou(boxes1, boxes2):
    return ops.box_iou(boxes1, boxes2)


def nms(boxes, scores, iou_threshold):
    return ops.nms(boxes, scores, iou_threshold)


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    return ops.nms(boxes, scores, iou_threshold, idxs=idxs)


def clip_boxes_to_image(boxes, size):
    h, w = size
    x = boxes[..., 0::2].clamp(min=0, max=w)
    y = boxes[..., 1::2].clamp(min=0, max=h)
    return torch.stack((x, y), dim=boxes.dim()).reshape(boxes.shape)


def remove_small_boxes(boxes, min_size):
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    return torch.where((ws >= min_size) & (hs >= min_size))[0]
