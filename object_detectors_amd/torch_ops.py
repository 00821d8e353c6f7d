"""The detection kernels as PyTorch custom operators (`torch.ops.mi355det.*`), registered with `torch.library` so that the dispatcher, fake
tensors / `torch.compile` shape inference and `torch.autograd` see them (north star: "surfaced to Python as PyTorch-ROCm custom ops through a
thin C ABI").  Every operator is a thin wrapper over the same `ops.*` ctypes call the module mirrors use; there is a "cuda" (= ROCm)
implementation only - calling one with CPU tensors raises NotImplementedError from the dispatcher, like the rest of the package there is no
CPU fallback.

    import object_detectors_amd.torch_ops            # registers the namespace
    iou  = torch.ops.mi355det.box_iou(b1, b2)        # torchvision.ops.box_iou           (tvision/_utils.py:271-344 call sites)
    keep = torch.ops.mi355det.nms(b, s, 0.5)         # torchvision.ops.nms
    keep = torch.ops.mi355det.batched_nms(b, s, idxs, 0.5)
    loss = torch.ops.mi355det.sigmoid_focal_loss_sum(x, t, 0.25, 2.0)      # differentiable in x (retinanet.py:137-141, reduction='sum')
    out  = torch.ops.mi355det.roi_align(feat, rois, 0.25, 7, 7, 2, False)  # differentiable in feat (roi_heads.py:818)
    iou  = torch.ops.mi355det.bbox_iou(bb1, bb2, 1, True)                  # yolo/utilities/helper.py:221-277 (IoU / GIoU / DIoU / CIoU)
"""
import torch
from torch.library import custom_op, register_autograd, register_fake

from . import ops

_DEV = "cuda"


# ---------------------------------------------------------------------------------------------- boxes
@custom_op("mi355det::box_iou", mutates_args=(), device_types=_DEV)
def box_iou(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    return ops.box_iou(boxes1, boxes2)


@register_fake("mi355det::box_iou")
def _(boxes1, boxes2):
    return boxes1.new_empty((boxes1.shape[0], boxes2.shape[0]), dtype=torch.float32)


@custom_op("mi355det::nms", mutates_args=(), device_types=_DEV)
def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    return ops.nms(boxes, scores, iou_threshold)


@register_fake("mi355det::nms")
def _(boxes, scores, iou_threshold):
    n = torch.library.get_ctx().new_dynamic_size()          # data-dependent number of kept boxes
    return boxes.new_empty((n,), dtype=torch.int64)


@custom_op("mi355det::batched_nms", mutates_args=(), device_types=_DEV)
def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    return ops.nms(boxes, scores, iou_threshold, idxs=idxs)


@register_fake("mi355det::batched_nms")
def _(boxes, scores, idxs, iou_threshold):
    n = torch.library.get_ctx().new_dynamic_size()
    return boxes.new_empty((n,), dtype=torch.int64)


@custom_op("mi355det::bbox_iou", mutates_args=(), device_types=_DEV)
def bbox_iou(bb1: torch.Tensor, bb2: torch.Tensor, iou_type: int, xcycwh: bool) -> torch.Tensor:
    return ops.bbox_iou(bb1, bb2, iou_type, xcycwh)


@register_fake("mi355det::bbox_iou")
def _(bb1, bb2, iou_type, xcycwh):
    shape = torch.broadcast_shapes(bb1.shape[:-1], bb2.shape[:-1])
    return bb1.new_empty(shape, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- focal loss (sum), fused forward + gradient
@custom_op("mi355det::sigmoid_focal_loss_fwd_bwd", mutates_args=(), device_types=_DEV)
def _focal_fwd_bwd(inputs: torch.Tensor, targets: torch.Tensor, alpha: float, gamma: float) -> tuple[torch.Tensor, torch.Tensor]:
    x = inputs.reshape(-1, inputs.shape[-1]) if inputs.dim() > 1 else inputs
    t = targets.reshape(-1, targets.shape[-1]) if targets.dim() > 1 else targets
    loss, grad = ops.sigmoid_focal_loss_sum(x, t, alpha, gamma, want_grad=True)
    return loss.reshape(()), grad.reshape(inputs.shape)


@register_fake("mi355det::sigmoid_focal_loss_fwd_bwd")
def _(inputs, targets, alpha, gamma):
    return inputs.new_empty((), dtype=torch.float32), inputs.new_empty(inputs.shape, dtype=torch.float32)


@custom_op("mi355det::sigmoid_focal_loss_sum", mutates_args=(), device_types=_DEV)
def sigmoid_focal_loss_sum(inputs: torch.Tensor, targets: torch.Tensor, alpha: float, gamma: float) -> torch.Tensor:
    return _focal_fwd_bwd(inputs, targets, alpha, gamma)[0]


@register_fake("mi355det::sigmoid_focal_loss_sum")
def _(inputs, targets, alpha, gamma):
    return inputs.new_empty((), dtype=torch.float32)


def _focal_setup(ctx, inputs, output):
    x, t, alpha, gamma = inputs
    ctx.save_for_backward(x, t)
    ctx.alpha, ctx.gamma = alpha, gamma


def _focal_backward(ctx, g):
    x, t = ctx.saved_tensors
    grad = _focal_fwd_bwd(x, t, ctx.alpha, ctx.gamma)[1]      # the kernel produces loss and gradient in one pass; recomputed here
    return grad * g, None, None, None


register_autograd("mi355det::sigmoid_focal_loss_sum", _focal_backward, setup_context=_focal_setup)


# ---------------------------------------------------------------------------------------------- RoIAlign (single level, NCHW fp32)
@custom_op("mi355det::roi_align", mutates_args=(), device_types=_DEV)
def roi_align(input: torch.Tensor, rois: torch.Tensor, spatial_scale: float, pooled_height: int, pooled_width: int, sampling_ratio: int,
              aligned: bool) -> torch.Tensor:
    return ops.roi_align_multi([input], rois, (pooled_height, pooled_width), [spatial_scale], sampling_ratio, aligned, 0, 0)


@register_fake("mi355det::roi_align")
def _(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, aligned):
    return input.new_empty((rois.shape[0], input.shape[1], pooled_height, pooled_width), dtype=torch.float32)


@custom_op("mi355det::roi_align_backward", mutates_args=(), device_types=_DEV)
def roi_align_backward(grad: torch.Tensor, input: torch.Tensor, rois: torch.Tensor, spatial_scale: float, pooled_height: int, pooled_width: int,
                       sampling_ratio: int, aligned: bool) -> torch.Tensor:
    return ops.roi_align_multi([input], rois, (pooled_height, pooled_width), [spatial_scale], sampling_ratio, aligned, 0, 0, grad_out=grad)[0]


@register_fake("mi355det::roi_align_backward")
def _(grad, input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, aligned):
    return input.new_empty(input.shape, dtype=torch.float32)


def _roi_setup(ctx, inputs, output):
    input, rois, spatial_scale, ph, pw, sampling_ratio, aligned = inputs
    ctx.save_for_backward(input, rois)
    ctx.args = (spatial_scale, ph, pw, sampling_ratio, aligned)


def _roi_backward(ctx, g):
    input, rois = ctx.saved_tensors
    return (roi_align_backward(g.contiguous(), input, rois, *ctx.args),) + (None,) * 6


register_autograd("mi355det::roi_align", _roi_backward, setup_context=_roi_setup)

OPS = ("box_iou", "nms", "batched_nms", "bbox_iou", "sigmoid_focal_loss_sum", "sigmoid_focal_loss_fwd_bwd", "roi_align", "roi_align_backward")
