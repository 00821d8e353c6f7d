"""`nn.Linear` of the Faster R-CNN box head (TwoMLPHead / FastRCNNPredictor, tvision/frcnn.py:243-289) on the library's MFMA kernels.

y = relu?(x @ W^T + b) is the 1x1 convolution of a [1, 1, R, in] NHWC tensor: forward = mi355det_conv_fwd_ex (bias and ReLU in the
epilogue), input gradient = mi355det_conv_dgrad, weight / bias gradient = mi355det_conv_wgrad - bf16 operands, fp32 accumulation, like
every other contraction of the engines.  Parameters stay ordinary fp32 torch parameters with nn.Linear's names and shapes (state_dict
compatible with the reference; `ParamGradSync` / torch optimizers see them unchanged); the bf16 packs are rebuilt when a parameter's
version counter moves (every optimizer step in training, never during an evaluation loop).
Output channels that are not a multiple of 32 (cls_score: K, bbox_pred: 4K) are zero-padded inside (the data-gradient GEMM reduces over them)."""
import math

import torch
from torch import nn

from .. import ops


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, weight, bias):
        R = x.shape[0]
        shape = ops.conv_shape(1, 1, R, mod.in_features, mod.out_store, 1, 1)
        wf, wd = mod._packs()
        xb = x.detach().to(torch.bfloat16).contiguous()
        if mod.relu:
            y = torch.empty((R, mod.out_store), device=x.device, dtype=torch.bfloat16)
            ops.conv_fwd_ex(shape, xb, wf, y, shift=mod._bias_store(), relu=True)
        else:
            y = torch.empty((R, mod.out_store), device=x.device, dtype=torch.float32)
            ops.conv_fwd_ex(shape, xb, wf, y, shift=mod._bias_store(), out_f32=True)
        ctx.mod, ctx.shape = mod, shape
        ctx.save_for_backward(xb, y if mod.relu else None, wd)
        return y[:, :mod.out_features]

    @staticmethod
    def backward(ctx, g):
        mod, shape = ctx.mod, ctx.shape
        xb, y, wd = ctx.saved_tensors
        R = xb.shape[0]
        dy = torch.zeros((R, mod.out_store), device=g.device, dtype=torch.bfloat16)
        if mod.relu:
            dy[:, :mod.out_features] = torch.where(y[:, :mod.out_features] > 0, g.to(torch.bfloat16), torch.zeros((), device=g.device, dtype=torch.bfloat16))
        else:
            dy[:, :mod.out_features] = g
        dx = None
        if ctx.needs_input_grad[1]:
            dx = torch.empty((R, mod.in_features), device=g.device, dtype=torch.bfloat16)
            ops.conv_dgrad(shape, dy, wd, dx)
        dw = torch.zeros((mod.out_store, mod.in_features), device=g.device, dtype=torch.float32)
        db = torch.zeros(mod.out_store, device=g.device, dtype=torch.float32)
        ops.conv_wgrad(shape, xb, dy, dw, dbias=db)
        return None, dx, dw[:mod.out_features], db[:mod.out_features]


class MfmaLinear(nn.Module):
    """Drop-in for nn.Linear(in_features, out_features) (+ optional fused ReLU) with nn.Linear's parameter names and initialisation."""

    def __init__(self, in_features, out_features, relu=False):
        super().__init__()
        if in_features % 64:
            raise ValueError("MfmaLinear: in_features must be a multiple of 64")
        self.in_features, self.out_features, self.relu = in_features, out_features, relu
        self.out_store = ops.pad_to(out_features, 32)
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)
        self._pack_key = None
        self._pack = None

    def _packs(self):
        w = self.weight
        key = (w._version, w.data_ptr(), self.bias._version)
        if key != self._pack_key:
            ws = w.detach()
            if self.out_store != self.out_features:
                ws = torch.cat([ws, torch.zeros((self.out_store - self.out_features, self.in_features), device=w.device, dtype=w.dtype)])
            shape = ops.conv_shape(1, 1, 8, self.in_features, self.out_store, 1, 1)
            wf, wd = ops.pack_weights(shape, ws.reshape(self.out_store, self.in_features, 1, 1))
            b = self.bias.detach()
            if self.out_store != self.out_features:
                b = torch.cat([b, torch.zeros(self.out_store - self.out_features, device=b.device, dtype=b.dtype)])
            self._pack, self._pack_key = (wf, wd, b.float().contiguous()), key
        return self._pack[0], self._pack[1]

    def _bias_store(self):
        return self._pack[2]

    def forward(self, x):
        if not x.is_cuda:
            raise ValueError("MfmaLinear: CUDA tensors only (no CPU fallback)")
        return _LinearFn.apply(self, x.reshape(x.shape[0], -1), self.weight, self.bias)
