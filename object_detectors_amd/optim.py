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
    def step(self, closure=None, grad_scale=1.0, zero_grad=False, skip_flag=None):
        """skip_flag: device int32 tensor; non-zero = leave parameters and momentum untouched (amp overflow, see DynamicLossScaler)."""
        g = self.param_groups[0]
        st = lib().mi355det_sgd_step_guarded(self.flat_w.data_ptr(), self.flat_g.data_ptr(), self.momentum_buf.data_ptr(), self.flat_w.numel(),
                                             float(g["lr"]), float(g["momentum"]), float(g["dampening"]), float(g["weight_decay"]),
                                             float(grad_scale), int(bool(g["nesterov"])), int(self.steps == 0), int(zero_grad),
                                             None if skip_flag is None else skip_flag.data_ptr(), _stream())
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
    def step(self, closure=None, grad_scale=1.0, zero_grad=False, skip_flag=None):
        g = self.param_groups[0]
        self.steps += 1
        st = lib().mi355det_adam_step_guarded(self.flat_w.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                              self.flat_w.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                              float(g["weight_decay"]), float(grad_scale), self.steps, int(zero_grad),
                                              None if skip_flag is None else skip_flag.data_ptr(), _stream())
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


class DynamicLossScaler:
    """apex amp's dynamic loss scaling around the optimizer step (yolo/procedures/initialize.py:44-45 `amp.initialize(..., opt_level)`,
    train_one_epoch.py:88-96 `with amp.scale_loss(loss, optimizer) as scaled_loss: scaled_loss.backward(); optimizer.step()`), on the
    flat buffers:

        scaler = DynamicLossScaler()
        out12 = engine.train_step(imgs, targets, criterion, grad_scale=scaler.loss_scale)     # gradients of loss * scale
        sync.wait()                                                                           # (multi-GPU: after the all-reduce)
        scaler.step(optimizer)                                                                # unscale + inf/nan check + guarded step + update

    apex's defaults: initial scale 2^16, x2 after 2000 clean steps, /2 (and the step skipped) on overflow, ceiling 2^24.  The overflow
    check and the skip run on the device (`mi355det_grad_nonfinite`, `mi355det_*_step_guarded`); `step` then reads the 4-byte flag to
    update the scale, the same host synchronisation apex performs.  With the engine's bf16 activations the scale is not needed for range
    (bf16 has fp32's exponent); the class exists so that a reference recipe that enables amp keeps its semantics (skipped steps included)."""

    def __init__(self, init_scale=2.0 ** 16, scale_factor=2.0, scale_window=2000, min_loss_scale=None, max_loss_scale=2.0 ** 24, enabled=True):
        self.loss_scale = float(init_scale) if enabled else 1.0
        self.scale_factor, self.scale_window = float(scale_factor), int(scale_window)
        self.min_loss_scale, self.max_loss_scale = min_loss_scale, max_loss_scale
        self.enabled = enabled
        self.unskipped = 0
        self.skipped_steps = 0
        self._flag = None

    def step(self, optimizer, zero_grad=False):
        """-> True when the step was applied, False when it was skipped for an inf / nan gradient."""
        if not self.enabled:
            optimizer.step(zero_grad=zero_grad)
            return True
        g = optimizer.flat_g
        if self._flag is None or self._flag.device != g.device:
            self._flag = torch.zeros(1, dtype=torch.int32, device=g.device)
        self._flag.zero_()
        check(lib().mi355det_grad_nonfinite(g.data_ptr(), g.numel(), self._flag.data_ptr(), _stream()), "grad_nonfinite")
        optimizer.step(grad_scale=1.0 / self.loss_scale, zero_grad=zero_grad, skip_flag=self._flag)
        overflow = bool(self._flag.item())
        if overflow:
            optimizer.steps -= 1                 # the guarded kernel left parameters and state as they were
            self.skipped_steps += 1
            self.loss_scale = self.loss_scale / self.scale_factor
            if self.min_loss_scale is not None:
                self.loss_scale = max(self.loss_scale, self.min_loss_scale)
            self.unskipped = 0
        else:
            self.unskipped += 1
            if self.unskipped == self.scale_window:
                self.loss_scale = min(self.max_loss_scale, self.loss_scale * self.scale_factor)
                self.unskipped = 0
        return not overflow

    def state_dict(self):
        return {"loss_scale": self.loss_scale, "unskipped": self.unskipped}

    def load_state_dict(self, sd):
        self.loss_scale, self.unskipped = float(sd["loss_scale"]), int(sd["unskipped"])
