"""Mirror of yolo/nets/yolohead.py:YoloHead (+ the DarkNet backbone it wraps): same constructor
config, same `forward(x) -> (out0, out1, out2)` with [bs, A*(5+C), H/32|16|8, W/32|16|8] outputs, and a
state_dict with the reference's keys/layouts — executed by the MI355X engine (engine.py)."""
import torch
import torch.nn as nn

from .engine import YoloV3Engine


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        ctx.mod = mod
        outs = mod.engine.forward(x, training=mod.training)
        return tuple(o for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        mod = ctx.mod
        mod.engine.backward([g for g in gouts])
        mod._attach_grads()
        return (None, None) + (None,) * len(mod._plist)


class YoloHead(nn.Module):
    def __init__(self, config, is_training=True):
        super().__init__()
        self.config = config
        cfgb = config["backbone"]
        na = len(config["dataset"]["anchors"][0])
        nc = config["yolo"]["classes"]
        # `batch_norm_sync` is the reference's switch for apex SyncBN (yolo/procedures/initialize.py:31-32; yolo/hydra/config.yaml)
        sync = bool(config.get("batch_norm_sync", False)) if hasattr(config, "get") else bool(getattr(config, "batch_norm_sync", False))
        # `apex_opt` (yolo/hydra/config.yaml:6, initialize.py:44-45): O1 / O2 / O3 store fp16 -> the engine's fp16 storage; O0 (fp32 in the
        # reference) -> the engine's default bf16 storage.  `storage` ("bf16" / "fp16") in the config overrides the mapping.
        opt = (config.get("apex_opt", "O0") if hasattr(config, "get") else getattr(config, "apex_opt", "O0")) or "O0"
        storage = (config.get("storage", None) if hasattr(config, "get") else getattr(config, "storage", None)) or ("fp16" if str(opt).upper() in ("O1", "O2", "O3") else "bf16")
        self.engine = YoloV3Engine(cfgb.get("backbone_name", "darknet_53"), na, nc, sync_bn=sync, storage=storage)
        self.layers_out_filters = [64, 128, 256, 512, 1024]
        # parameters are views into the engine's flat master buffer (conv weights in OHWI layout)
        self._pnames, self._plist = [], nn.ParameterList()
        for name, _o, _n, _shape in self.engine.param_order:
            self._pnames.append(name)
            self._plist.append(nn.Parameter(self.engine.params[name]))
        self.train(is_training)

    def train(self, mode=True):
        """model.eval() (test_one_epoch.py:10, valid_one_epoch.py:9) declares the weights static until the next train() / load: the engine
        then packs them and folds the BatchNorm layers once per evaluation loop instead of once per batch."""
        super().train(mode)
        if hasattr(self, "engine"):
            self.engine.freeze_inference(not mode)
        return self

    def _attach_grads(self):
        for name, p in zip(self._pnames, self._plist):
            p.grad = self.engine.grads[name]

    def named_reference_parameters(self):
        return list(zip(self._pnames, self._plist))

    def forward(self, x):
        if torch.is_grad_enabled() and self.training:
            return _EngineFn.apply(self, x, *self._plist)
        return tuple(self.engine.forward(x, training=self.training))

    # reference-compatible checkpoints (initialize.py:12-25,57-104 save/load model.state_dict())
    def state_dict(self, *a, **k):
        return self.engine.reference_state_dict()

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.engine.load_reference_state_dict(sd)

    def load_darknet_weights(self, weights_path):
        """yolohead.py:90-165: a darknet `.weights` file = 5 int32 header words, then float32 values consumed while walking the
        state dict: for every conv + BN pair `bn.bias, bn.weight, bn.running_mean, bn.running_var, conv.weight`, for a biased conv
        (the three conv_out layers) `conv.bias, conv.weight`.  (The reference's walk raises on `num_batches_tracked`, a key that
        did not exist in the torch it was written for; it is skipped here.)  Returns the number of floats consumed."""
        import numpy as np
        with open(weights_path, "rb") as fp:
            np.fromfile(fp, dtype=np.int32, count=5)
            weights = np.fromfile(fp, dtype=np.float32)
        sd = self.state_dict()
        ptr = 0
        last_bn_weight = last_conv = None

        def take(key):
            nonlocal ptr
            n = sd[key].numel()
            if ptr + n > weights.size:
                raise ValueError(f"{weights_path}: file ends inside {key} ({weights.size} floats)")
            sd[key] = torch.from_numpy(weights[ptr:ptr + n].copy()).view_as(sd[key])
            ptr += n
        for k in list(sd.keys()):
            if "bn" in k:
                if "num_batches_tracked" in k:
                    continue
                if "weight" in k:
                    last_bn_weight = k
                elif "bias" in k:
                    take(k)
                    take(last_bn_weight)
                    last_bn_weight = None
                elif "running_mean" in k:
                    take(k)
                elif "running_var" in k:
                    take(k)
                    take(last_conv)
                    last_conv = None
                else:
                    raise Exception("Error for bn")
            elif "conv" in k:
                if "weight" in k:
                    last_conv = k
                else:
                    take(k)
                    take(last_conv)
                    last_conv = None
        self.load_state_dict(sd)
        return ptr
