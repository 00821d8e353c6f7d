"""Mirror of the multi-scale part of yolo/procedures/train_one_epoch.py (:15-26 get_new_scale, :64-69 the resize): every
`multiscaler.freq` iterations rank 0 draws a new input size (a multiple of 32 inside bounds x inp_dim), broadcasts it, and the batch is
resized with bilinear interpolation before the forward pass.  The resize runs on mi355det_resize_bilinear."""
import math
import random

import torch
import torch.distributed as dist

from ...tvision.transform import interpolate_bilinear


def _get(cfg, *names):
    for n in names:
        cfg = cfg[n] if isinstance(cfg, dict) else getattr(cfg, n)
    return cfg


def get_new_scale(cfg, device="cuda"):
    """train_one_epoch.py:15-26 (same `random.randrange` draw; the rank-0 value wins when `multiscaler.broadcast`)."""
    imgsz = _get(cfg, "dataset", "inp_dim")
    bounds = _get(cfg, "multiscaler", "bounds")
    lb = math.ceil(imgsz * bounds[0] / 32)
    ub = math.floor(imgsz * bounds[1] / 32)
    sf = torch.tensor([random.randrange(lb, ub)], device=device)
    if _get(cfg, "multiscaler", "broadcast") is True and dist.is_available() and dist.is_initialized():
        dist.broadcast(sf, 0)
    return sf.item() * 32


def multiscale_batch(imgs, new_scale, inp_dim):
    """train_one_epoch.py:68-69: F.interpolate(imgs, size=new_scale, mode='bilinear', align_corners=False) when the scale differs."""
    if new_scale == inp_dim:
        return imgs
    return interpolate_bilinear(imgs, new_scale)
