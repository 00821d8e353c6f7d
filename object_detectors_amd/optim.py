"""Fused optimizer steps on the engine's flat fp32 parameter / gradient buffers.

Replaces `optim.SGD(model.parameters(), lr, momentum, weight_decay)` / `optim.Adam(...)` of
yolo/procedures/initialize.py:38,41 and the `optimizer.step()` / `optimizer.zero_grad()` pair of
yolo/procedures/train_one_epoch.py:96 with ONE kernel over {param, grad, state}
(`mi355det_sgd_step` / `mi355det_adam_step`).  Both classes derive from `torch.optim.Optimizer` so the
reference's `optim.lr_scheduler.*` objects (initialize.py:110-126) drive `param_groups[0]['lr']` unchanged.
"""
import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Flat(torch.optim.Optimizer):
    def __init__(self, flat_w, flat_g, defaults):
        if flat_w.dtype != torch.float32 or flat_g.dtype != torch.float32 or flat_w.numel() != flat_g.numel():
            raise ValueError("flat parameter and gradient buffers must be fp32 and equally sized")
        if not flat_w.is_cuda:
            raise ValueError("mi355det optimizers run on the GPU buffers of an engine (no CPU path)")
        self.flat_w, self.flat_g = flat_w, flat_g
        super().__init__([flat_w], defaults)
        self.steps = 0

    @classmethod
    def for_engine(cls, engine, **kw):
        return cls(engine.flat_w, engine.flat_g, **kw)

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()


class FlatSGD(_Flat):
    """torch.optim.SGD semantics (weight decay added to the gradient, momentum buffer seeded with the first gradient)."""

    def __init__(self, flat_w, flat_g, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        super().__init__(flat_w, flat_g, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))
        self.momentum_buf = torch.zeros_like(flat_w)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, zero_grad=False):
        g = self.param_groups[0]
        st = lib().mi355det_sgd_step(self.flat_w.data_ptr(), self.flat_g.data_ptr(), self.momentum_buf.data_ptr(), self.flat_w.numel(),
                                     float(g["lr"]), float(g["momentum"]), float(g["dampening"]), float(g["weight_decay"]), float(grad_scale),
                                     int(bool(g["nesterov"])), int(self.steps == 0), int(zero_grad), _stream())
        check(st, "sgd_step")
        self.steps += 1


    # ---- the reference's checkpoint format (initialize.py:18-25 saves optimizer.state_dict() of torch.optim.SGD over model.parameters())
    def reference_state_dict(self, engine):
        g = self.param_groups[0]
        tensors = engine.reference_parameter_tensors(self.momentum_buf)
        state = {i: {"momentum_buffer": t} for i, t in enumerate(tensors)} if self.steps > 0 and g["momentum"] != 0 else {}
        group = {k: g[k] for k in ("lr", "momentum", "dampening", "weight_decay", "nesterov")}
        group["params"] = list(range(len(tensors)))
        return {"state": state, "param_groups": [group]}

    def load_reference_state_dict(self, engine, sd):
        g = sd["param_groups"][0]
        for k in ("lr", "momentum", "dampening", "weight_decay", "nesterov"):
            if k in g:
                self.param_groups[0][k] = g[k]
        if sd["state"]:
            n = len(g["params"])
            bufs = [sd["state"][i]["momentum_buffer"] for i in range(n)]
            engine.load_reference_parameter_tensors(bufs, self.momentum_buf)
            self.steps = max(self.steps, 1)
        else:
            self.momentum_buf.zero_()
            self.steps = 0


class FlatAdam(_Flat):
    """torch.optim.Adam semantics (L2 weight decay added to the gradient, bias-corrected moments)."""

    def __init__(self, flat_w, flat_g, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(flat_w, flat_g, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(flat_w)
        self.exp_avg_sq = torch.zeros_like(flat_w)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, zero_grad=False):
        g = self.param_groups[0]
        self.steps += 1
        st = lib().mi355det_adam_step(self.flat_w.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                      self.flat_w.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                      float(g["weight_decay"]), float(grad_scale), self.steps, int(zero_grad), _stream())
        check(st, "adam_step")

    def reference_state_dict(self, engine):
        g = self.param_groups[0]
        m, v = engine.reference_parameter_tensors(self.exp_avg), engine.reference_parameter_tensors(self.exp_avg_sq)
        state = {i: {"step": torch.tensor(float(self.steps)), "exp_avg": a, "exp_avg_sq": b} for i, (a, b) in enumerate(zip(m, v))} if self.steps else {}
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
                 "params": list(range(len(m)))}
        return {"state": state, "param_groups": [group]}

    def load_reference_state_dict(self, engine, sd):
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in g:
                self.param_groups[0][k] = g[k]
        if sd["state"]:
            n = len(g["params"])
            engine.load_reference_parameter_tensors([sd["state"][i]["exp_avg"] for i in range(n)], self.exp_avg)
            engine.load_reference_parameter_tensors([sd["state"][i]["exp_avg_sq"] for i in range(n)], self.exp_avg_sq)
            self.steps = int(sd["state"][0]["step"])
        else:
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            self.steps = 0
