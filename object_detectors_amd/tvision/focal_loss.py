"""Mirror of torchvision.ops.sigmoid_focal_loss (call site tvision/retinanet.py:137-141) and the
RetinaNet classification loss (retinanet.py:107-143) fused forward+backward in HIP."""
import torch

from .. import ops


class _FocalSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, alpha, gamma):
        loss, grad = ops.sigmoid_focal_loss_sum(inputs.reshape(-1, inputs.shape[-1]) if inputs.dim() > 1 else inputs,
                                                targets.reshape(-1, targets.shape[-1]) if targets.dim() > 1 else targets,
                                                alpha, gamma, want_grad=inputs.requires_grad)
        ctx.grad = None if grad is None else grad.reshape(inputs.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.grad is None else ctx.grad * g), None, None, None


class _FocalElem(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, alpha, gamma):
        loss, grad = ops.sigmoid_focal_loss_elem(inputs, targets, alpha, gamma, want_grad=inputs.requires_grad)
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.grad is None else ctx.grad * g), None, None, None


def sigmoid_focal_loss(inputs, targets, alpha=0.25, gamma=2, reduction="none"):
    """torchvision.ops.sigmoid_focal_loss: 'none' (torchvision's default: elementwise loss), 'mean', 'sum' (what the reference calls)."""
    if reduction == "sum":
        return _FocalSum.apply(inputs, targets, alpha, gamma)
    if reduction == "mean":
        return _FocalSum.apply(inputs, targets, alpha, gamma) / inputs.numel()
    if reduction == "none":
        return _FocalElem.apply(inputs, targets, alpha, gamma)
    raise ValueError(f"Invalid Value for arg 'reduction': '{reduction}' \n Supported reduction modes: 'none', 'mean', 'sum'")


class _RetinaCls(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, matched, gt_labels, tfidf, alpha, gamma):
        loss, grad = ops.retina_cls_loss_sum(logits, matched, gt_labels, alpha, gamma, scale=tfidf,
                                             want_grad=logits.requires_grad)
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.grad is None else ctx.grad * g), None, None, None, None, None


def retinanet_classification_loss(cls_logits, targets, matched_idxs, tfidf=None, alpha=0.25, gamma=2.0):
    """RetinaNetClassificationHead.compute_loss (retinanet.py:107-143) without the dense one-hot target."""
    losses = []
    for t, logits, mi in zip(targets, cls_logits, matched_idxs):
        nfg = (mi >= 0).sum().clamp(min=1)
        losses.append(_RetinaCls.apply(logits, mi, t["labels"], tfidf, alpha, gamma) / nfg)
    return sum(losses) / len(targets)
