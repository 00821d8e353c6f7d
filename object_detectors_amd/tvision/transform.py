"""Mirror of `GeneralizedRCNNTransform` (torchvision_models/tvision/transform.py:65-257) and `resize_boxes` (:279-293): the input side of the
RetinaNet / Faster R-CNN step.

    transform = GeneralizedRCNNTransform(800, 1333, image_mean, image_std)
    image_list, targets = transform(images, targets)      # images: list of [3,H,W] float tensors on the GPU, any sizes
    image_list.tensors   [N,3,Hp,Wp] normalised, bilinear-resized (min side 800 / max side 1333), zero-padded to a multiple of 32
    image_list.image_sizes   [(h,w)] of every resized image
    detections = transform.postprocess(detections, image_list.image_sizes, original_image_sizes)   # boxes back to the input frame

Normalise + resize + pad run as ONE kernel per image (mi355det_resize_bilinear) that writes straight into the image's slot of the padded
batch; the size arithmetic (float32 scale, floor of the double product) is host logic that follows the reference line by line so the
output SIZES are identical."""
import math

import numpy as np
import torch

from .._lib import check, lib, ptr, stream_ptr


class ImageList:
    """tvision/image_list.py:7-27."""

    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device):
        return ImageList(self.tensors.to(device), self.image_sizes)


def resized_size(h, w, self_min_size, self_max_size):
    """Output size of `_resize_image_and_masks` (transform.py:26-52): scale = min(min_size / min(h,w), max_size / max(h,w)) in float32, then
    F.interpolate(scale_factor=scale.item(), recompute_scale_factor=True) -> floor(float(size) * scale) in double."""
    # `self_min_size / min_size` in the reference has a Python float on the left and a float32 tensor on the right = Tensor.__rtruediv__ =
    # reciprocal() * scalar: two float32 roundings (1/480 * 800 = 1.6666667, not 1.6666666), which decides 800 vs 799 rows
    mn, mx = np.float32(min(h, w)), np.float32(max(h, w))
    scale = float(min((np.float32(1) / mn) * np.float32(self_min_size), (np.float32(1) / mx) * np.float32(self_max_size)))
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))


def resize_boxes(boxes, original_size, new_size):
    """transform.py:279-293."""
    if boxes.numel() == 0:
        return boxes.clone()
    b = boxes.float().contiguous()
    out = torch.empty_like(b)
    check(lib().mi355det_resize_boxes(ptr(b), ptr(out), b.shape[0], int(original_size[0]), int(original_size[1]), int(new_size[0]), int(new_size[1]),
                                      stream_ptr()), "resize_boxes")
    return out


class GeneralizedRCNNTransform(torch.nn.Module):
    def __init__(self, min_size, max_size, image_mean, image_std, size_divisible=32, fixed_size=None):
        super().__init__()
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size
        self.image_mean, self.image_std = image_mean, image_std
        self.size_divisible = size_divisible
        self.fixed_size = fixed_size
        self._stats = {}

    def torch_choice(self, k):
        """transform.py:137-145 (same RNG draw as the reference)."""
        return k[int(torch.empty(1).uniform_(0.0, float(len(k))).item())]

    def _mean_std(self, device):
        if device not in self._stats:
            self._stats[device] = (torch.tensor(self.image_mean, dtype=torch.float32, device=device),
                                   torch.tensor(self.image_std, dtype=torch.float32, device=device))
        return self._stats[device]

    def forward(self, images, targets=None):
        images = list(images)
        if targets is not None:
            targets = [dict(t) for t in targets]          # copy, as the reference does, so the caller's dicts are not modified
        sizes = []
        for img in images:
            if img.dim() != 3:
                raise ValueError("images is expected to be a list of 3d tensors of shape [C, H, W], got {}".format(img.shape))
            if not img.is_floating_point():
                raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {img.dtype} instead")
            h, w = int(img.shape[-2]), int(img.shape[-1])
            if self.fixed_size is not None:
                sizes.append((int(self.fixed_size[1]), int(self.fixed_size[0])))
            else:
                size = float(self.torch_choice(self.min_size)) if self.training else float(self.min_size[-1])
                sizes.append(resized_size(h, w, size, float(self.max_size)))
        stride = float(self.size_divisible)
        ph = int(math.ceil(float(max(s[0] for s in sizes)) / stride) * stride)
        pw = int(math.ceil(float(max(s[1] for s in sizes)) / stride) * stride)
        dev = images[0].device
        c = int(images[0].shape[0])
        mean, std = self._mean_std(dev)
        batch = torch.empty((len(images), c, ph, pw), dtype=torch.float32, device=dev)
        L = lib()
        for i, (img, (oh, ow)) in enumerate(zip(images, sizes)):
            src = img.float().contiguous()
            check(L.mi355det_resize_bilinear(ptr(src), c, c, int(img.shape[-2]), int(img.shape[-1]), ptr(mean), ptr(std), ptr(batch[i]), oh, ow, ph, pw,
                                             stream_ptr()), "resize_bilinear")
            if targets is not None and targets[i] is not None and "boxes" in targets[i]:
                targets[i]["boxes"] = resize_boxes(targets[i]["boxes"], (int(img.shape[-2]), int(img.shape[-1])), (oh, ow))
        return ImageList(batch, sizes), targets

    def postprocess(self, result, image_shapes, original_image_sizes):
        """transform.py:228-247 (boxes only: no mask / keypoint branch on this path)."""
        if self.training:
            return result
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            result[i]["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
        return result


def interpolate_bilinear(imgs, size):
    """F.interpolate(imgs, size=size, mode='bilinear', align_corners=False) of a batch [N,C,H,W] (YOLO multi-scale training,
    yolo/procedures/train_one_epoch.py:64-69)."""
    n, c, h, w = imgs.shape
    oh, ow = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
    src = imgs.float().contiguous()
    out = torch.empty((n, c, oh, ow), dtype=torch.float32, device=imgs.device)
    check(lib().mi355det_resize_bilinear(ptr(src), n * c, c, h, w, None, None, ptr(out), oh, ow, oh, ow, stream_ptr()), "resize_bilinear")
    return out
