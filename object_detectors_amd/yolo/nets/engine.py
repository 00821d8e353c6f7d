"""YOLOv3 (Darknet backbone + YoloHead) executor over libmi355det.so.

Mirrors the forward graph of the reference's `DarkNet` (yolo/nets/backbone/darknet.py:37-84) and
`YoloHead` (yolo/nets/yolohead.py:14-88) and owns the matching backward.  MI355X-first layout:

  * activations NHWC bf16, resident in HBM for the whole step (z = pre-BN conv output and a = activated
    output are both kept: ~12 GB at bs 32 / 640 px, trivial against 288 GB);
  * channel concat is free: the producer writes straight into its channel slice of the concat buffer;
  * every op is one prepared C-ABI call; a step is a static list of (function, ctypes args) tuples built
    once per input shape ("plan"), so the host only walks a list and the GPU queue stays full;
  * master weights / gradients live in two flat fp32 buffers (one memset, bucketed all-reduce, fused
    optimizers); conv weights are OHWI so the weight-gradient GEMM output IS the parameter gradient.
"""
import ctypes as C
import math
import os

import torch

from ... import _lib, ops, tune
from ..._lib import check, lib

BLOCKS = {"darknet_21": [1, 1, 2, 2, 1], "darknet_53": [1, 2, 8, 8, 4]}
SLOPE = 0.1
BN_EPS, BN_MOM = 1e-5, 0.1


def _vp(t, byte_off=0):
    return C.c_void_p(t.data_ptr() + byte_off) if t is not None else None


class Act:
    """A [n,h,w,c] bf16 activation living in (a channel slice of) an NHWC buffer."""

    def __init__(self, buf, n, h, w, c, ld, ch_off=0):
        self.buf, self.n, self.h, self.w, self.c, self.ld, self.ch_off = buf, n, h, w, c, ld, ch_off
        self.grad = None
        self.grad_written = False
        self.skips = []
        self.conv_consumers = 0      # convolutions reading this activation (their dgrads all add into its gradient)

    @property
    def ptr(self):
        return C.c_void_p(self.buf.data_ptr() + 2 * self.ch_off)

    @property
    def pixels(self):
        return self.n * self.h * self.w

    def slice(self, c0, c):
        return Act(self.buf, self.n, self.h, self.w, c, self.ld, self.ch_off + c0)


class ConvSpec:
    def __init__(self, name, cin, cout, k, stride, bn=True, bias=False):
        self.name, self.cin, self.cout, self.k, self.stride, self.bn, self.bias = name, cin, cout, k, stride, bn, bias


def arch(backbone="darknet_53", na=3, nc=80):
    """Ordered conv specs (reference state_dict order) keyed by name."""
    specs = [ConvSpec("backbone.conv1", 3, 32, 3, 1)]
    inpl = 32
    for li, (planes, nb) in enumerate(zip([(32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)], BLOCKS[backbone]), 1):
        p = f"backbone.layer{li}"
        specs.append(ConvSpec(p + ".ds_conv", inpl, planes[1], 3, 2))
        inpl = planes[1]
        for b in range(nb):
            specs.append(ConvSpec(f"{p}.residual_{b}.conv1", inpl, planes[0], 1, 1))
            specs.append(ConvSpec(f"{p}.residual_{b}.conv2", planes[0], inpl, 3, 1))
    fo = na * (5 + nc)

    def emb(name, fl, cin):
        ch = [(cin, fl[0], 1), (fl[0], fl[1], 3), (fl[1], fl[0], 1), (fl[0], fl[1], 3), (fl[1], fl[0], 1), (fl[0], fl[1], 3)]
        for i, (ci, co, k) in enumerate(ch):
            specs.append(ConvSpec(f"{name}.{i}.conv", ci, co, k, 1))
        specs.append(ConvSpec(name + ".conv_out", fl[1], fo, 1, 1, bn=False, bias=True))
    emb("embedding0", (512, 1024), 1024)
    specs.append(ConvSpec("embedding1_cbl.conv", 512, 256, 1, 1))
    emb("embedding1", (256, 512), 768)
    specs.append(ConvSpec("embedding2_cbl.conv", 256, 128, 1, 1))
    emb("embedding2", (128, 256), 384)
    return specs


def bn_name(conv_name):
    """reference BatchNorm module name next to a conv (darknet.py / yolohead.py naming)."""
    if conv_name == "backbone.conv1":
        return "backbone.bn1"
    if conv_name.endswith(".ds_conv"):
        return conv_name[:-len("ds_conv")] + "ds_bn"
    if conv_name.endswith(".conv1"):
        return conv_name[:-len("conv1")] + "bn1"
    if conv_name.endswith(".conv2"):
        return conv_name[:-len("conv2")] + "bn2"
    return conv_name[:-len("conv")] + "bn"


class YoloV3Engine:
    def __init__(self, backbone="darknet_53", num_anchors=3, num_classes=80, device=None, seed=0, sync_bn=False, process_group=None, storage="bf16",
                 deterministic=True, fuse_bn_reduce=False):
        """deterministic (default): every kernel of the step sums in a fixed order - two runs of the same inputs under the same tune record give
        bit-identical weights (tests/test_gpu_trajectory.py), like the reference's torch / cuDNN BatchNorm backward.  False: the BatchNorm-backward
        sums end in fp32 atomics instead of a second small launch per layer (72 launches on the dependency chain: ~0.25 ms of a 28 ms step,
        profiles/r04_ab_results.md); gradients then differ by ~6e-4 of max from run to run.
        fuse_bn_reduce (experimental): the BatchNorm-backward sums in the epilogue of the data gradient that produces the activation
        gradient (mi355det_conv_dgrad_bn) instead of a separate pass; correct and tested, 0.7 ms slower per step (the epilogue's z-tile reads are not
        prefetched), off by default.
        storage: format of every stored activation, activation gradient and packed weight - "bf16" (default) or "fp16", the format of the
        reference's mixed-precision recipe (apex O2: yolo/batch_files/sample.txt:28-44, initialize.py:44-45); accumulation, BatchNorm statistics,
        master weights and gradients stay fp32 either way.  fp16 has three more mantissa bits and five fewer exponent bits: train it with a loss
        scale (optim.DynamicLossScaler, train_step(..., grad_scale=S)) exactly as the reference does.  MI355DET_STORAGE overrides the default.
        sync_bn: the `batch_norm_sync` switch of the reference (yolo/procedures/initialize.py:31-32, apex convert_syncbn_model): every
        BatchNorm layer normalises with the statistics of the GLOBAL batch - one all-reduce of the per-channel (sum, sum of squares) in
        forward and one of (sum dy, sum dy*xhat) in backward per layer (2C floats each: latency-bound, 72 layers)."""
        lib()   # fail loudly if the HIP library is missing
        storage = os.environ.get("MI355DET_STORAGE", storage) if storage == "bf16" else storage
        if storage not in ("bf16", "fp16"):
            raise ValueError("storage must be 'bf16' or 'fp16'")
        self.storage = storage
        self.deterministic = bool(deterministic)
        self.fuse_bn_reduce = bool(fuse_bn_reduce)
        self.L = _lib.storage_lib(storage)                 # entry points of this storage format (fp16: the *_f16 twins)
        self.adt = torch.float16 if storage == "fp16" else torch.bfloat16
        self.grad_fmt = 2 if storage == "fp16" else 1       # mi355det_yolo_loss_cfg.grad_is_bf16
        self.sync_bn, self.process_group = bool(sync_bn), process_group
        if self.sync_bn and process_group is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                # the 2 x 72 small statistics all-reduces are BLOCKING steps of the forward / backward dependency chain: on their own
                # communicator they do not queue behind the asynchronous gradient buckets of the default group (every rank constructs
                # its engine at the same point, so this collective call is matched)
                self.process_group = dist.new_group()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.backbone, self.na, self.nc = backbone, num_anchors, num_classes
        self.head_c = num_anchors * (5 + num_classes)
        self.head_ld = ops.pad_to(self.head_c, 32)
        self._static_epoch = 1          # version of the parameters / buffers: strictly increasing, bumped by every change and every freeze / unfreeze
        self._frozen = False            # True while the weights are declared static (freeze_inference)
        self.specs = arch(backbone, num_anchors, num_classes)
        self.by_name = {s.name: s for s in self.specs}
        self._layout_params()
        self.reset_parameters(seed)
        self.plans = {}
        self.training = True
        self.num_batches_tracked = 0

    def freeze_inference(self, on=True):
        """Declare the parameters and BatchNorm buffers unchanged from now on (model.eval() before a test loop, test_one_epoch.py:10-16):
        eval-mode forwards then re-pack the bf16 weights and recompute the 72 folded BatchNorm scale / shift rows only ONCE instead of on
        every batch (0.2 ms + 72 launches).  Any load_* / reset / training forward un-freezes; code that writes `params` / `buffers`
        directly must call `freeze_inference(True)` again (or `False`)."""
        self._static_epoch += 1         # never reused: a plan whose constants were built for an earlier freeze cannot match again
        self._frozen = bool(on)

    def _weights_changed(self):
        self._static_epoch += 1

    # ------------------------------------------------------------------ parameters
    def _layout_params(self):
        """Flat fp32 parameter/gradient buffers; views per tensor.  Conv weights OHWI [cout,k,k,cin]
        (stem: [32, 32] im2col form, conv_out rows padded to head_ld)."""
        off = 0
        self.pviews = {}       # name -> (offset, numel, shape)
        order = []
        for s in self.specs:
            if s.name == "backbone.conv1":
                shape = (32, 32)
            elif not s.bn:
                shape = (ops.pad_to(s.cout, 32), s.k, s.k, s.cin)
            else:
                shape = (s.cout, s.k, s.k, s.cin)
            n = math.prod(shape)
            order.append((s.name + ".weight", off, n, shape))
            off += ops.pad_to(n, 64)
            if s.bias:
                order.append((s.name + ".bias", off, s.cout, (s.cout,)))
                off += ops.pad_to(s.cout, 64)
            if s.bn:
                b = bn_name(s.name)
                for suffix in (".weight", ".bias"):
                    order.append((b + suffix, off, s.cout, (s.cout,)))
                    off += ops.pad_to(s.cout, 64)
        self.flat_w = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.param_order = order
        self.params, self.grads = {}, {}
        for name, o, n, shape in order:
            self.params[name] = self.flat_w[o:o + n].view(shape)
            self.grads[name] = self.flat_g[o:o + n].view(shape)
        # BN running statistics
        self.buffers = {}
        for s in self.specs:
            if s.bn:
                b = bn_name(s.name)
                self.buffers[b + ".running_mean"] = torch.zeros(s.cout, device=self.device)
                self.buffers[b + ".running_var"] = torch.ones(s.cout, device=self.device)
        # packed bf16 weights
        self.packed = {}
        for s in self.specs:
            shp = self._wshape(s, 1, 8, 8)   # geometry-independent sizes
            cp = ops.cout_pad_of(shp.cout)
            wf = torch.zeros(cp * shp.ksize * shp.ksize * shp.cin, device=self.device, dtype=self.adt)
            wd = None
            if s.name != "backbone.conv1":
                wd = torch.zeros(self.L.mi355det_dgrad_pack_elems(C.byref(shp)), device=self.device, dtype=self.adt)
            self.packed[s.name] = (wf, wd)

    def _wshape(self, s, n, h, w, in_ld=None, out_ld=None):
        if s.name == "backbone.conv1":
            return ops.conv_shape(n, h, w, 32, 32, 1, 1, in_ld, out_ld)
        cout = s.cout if s.bn else ops.pad_to(s.cout, 32)
        return ops.conv_shape(n, h, w, s.cin, cout, s.k, s.stride, in_ld, out_ld)

    def reset_parameters(self, seed=0):
        """Reference initialisation: conv N(0, sqrt(2/(k*k*cout))) in the backbone (darknet.py:53-59), PyTorch
        default (kaiming-uniform a=sqrt(5)) for the head convs, BN weight 1 / bias 0."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        for s in self.specs:
            w = self.params[s.name + ".weight"]
            if s.name.startswith("backbone."):
                std = math.sqrt(2.0 / (s.k * s.k * s.cout))
                t = torch.randn((s.cout, s.cin, s.k, s.k), generator=g) * std
            else:
                bound = 1.0 / math.sqrt(s.cin * s.k * s.k)
                t = (torch.rand((s.cout, s.cin, s.k, s.k), generator=g) * 2 - 1) * bound
            self._set_weight_oihw(s, t)
            if s.bias:
                bound = 1.0 / math.sqrt(s.cin * s.k * s.k)
                self.params[s.name + ".bias"].copy_((torch.rand(s.cout, generator=g) * 2 - 1) * bound)
            if s.bn:
                b = bn_name(s.name)
                self.params[b + ".weight"].fill_(1.0)
                self.params[b + ".bias"].zero_()
            del w
        self._weights_changed()

    def flat_views(self, flat):
        """Per-tensor views (engine layout) of any flat buffer laid out like flat_w: gradients, optimizer state."""
        if flat.numel() != self.flat_w.numel():
            raise ValueError("buffer does not have the engine's flat layout")
        return {name: flat[o:o + n].view(shape) for name, o, n, shape in self.param_order}

    def _set_weight_oihw(self, s, t, dst=None):
        """Load a reference-layout [cout,cin,k,k] tensor into the engine layout."""
        w = (dst or self.params)[s.name + ".weight"]
        t = t.to(self.device, torch.float32)
        if s.name == "backbone.conv1":
            w.zero_()
            w[:, :27] = t.permute(0, 2, 3, 1).reshape(32, 27)      # k = (kh*3+kw)*3 + c
        elif not s.bn:
            w.zero_()
            w[:s.cout] = t.permute(0, 2, 3, 1)
        else:
            w.copy_(t.permute(0, 2, 3, 1))

    def _get_weight_oihw(self, s, src=None):
        w = (src or self.params)[s.name + ".weight"]
        if s.name == "backbone.conv1":
            return w[:, :27].reshape(32, 3, 3, 3).permute(0, 3, 1, 2).contiguous()
        if not s.bn:
            return w[:s.cout].permute(0, 3, 1, 2).contiguous()
        return w.permute(0, 3, 1, 2).contiguous()

    def load_reference_state_dict(self, sd):
        """state_dict of the reference YoloHead (keys as in darknet.py / yolohead.py)."""
        for s in self.specs:
            self._set_weight_oihw(s, sd[s.name + ".weight"])
            if s.bias:
                self.params[s.name + ".bias"].copy_(sd[s.name + ".bias"])
            if s.bn:
                b = bn_name(s.name)
                self.params[b + ".weight"].copy_(sd[b + ".weight"])
                self.params[b + ".bias"].copy_(sd[b + ".bias"])
                for k in (".running_mean", ".running_var"):
                    if b + k in sd:
                        self.buffers[b + k].copy_(sd[b + k])
                if b + ".num_batches_tracked" in sd:
                    self.num_batches_tracked = int(sd[b + ".num_batches_tracked"])
        self._weights_changed()

    def reference_parameter_tensors(self, flat):
        """A flat buffer with the engine's layout (e.g. an optimizer's momentum) as the list of tensors the reference's
        `model.parameters()` would pair with it: reference layouts (OIHW), reference order."""
        sd = self.reference_state_dict(src=self.flat_views(flat), params_only=True)
        return list(sd.values())

    def load_reference_parameter_tensors(self, tensors, flat):
        """Inverse of reference_parameter_tensors: fill `flat` from per-parameter tensors in the reference's parameters() order."""
        dst = self.flat_views(flat)
        names = list(self.reference_state_dict(params_only=True).keys())
        if len(names) != len(tensors):
            raise ValueError(f"expected {len(names)} parameter tensors, got {len(tensors)}")
        byname = dict(zip(names, tensors))
        for s in self.specs:
            self._set_weight_oihw(s, byname[s.name + ".weight"], dst)
            if s.bias:
                dst[s.name + ".bias"].copy_(byname[s.name + ".bias"])
            if s.bn:
                b = bn_name(s.name)
                dst[b + ".weight"].copy_(byname[b + ".weight"])
                dst[b + ".bias"].copy_(byname[b + ".bias"])
        self._weights_changed()

    def reference_state_dict(self, grads=False, src=None, params_only=False):
        """Parameters (or their gradients, or any buffer given as `src` views) under the reference's names, layouts and key order."""
        out = {}
        if src is None:
            src = self.grads if grads else self.params
        else:
            grads = True
        for s in self.specs:
            out[s.name + ".weight"] = self._get_weight_oihw(s, src)
            if s.bias:
                out[s.name + ".bias"] = src[s.name + ".bias"].clone()
            if s.bn:
                b = bn_name(s.name)
                out[b + ".weight"] = src[b + ".weight"].clone()
                out[b + ".bias"] = src[b + ".bias"].clone()
                if not grads and not params_only:
                    for k in (".running_mean", ".running_var"):
                        out[b + k] = self.buffers[b + k].clone()
                    out[b + ".num_batches_tracked"] = torch.tensor(self.num_batches_tracked, dtype=torch.int64)
        return out

    # ------------------------------------------------------------------ plan
    MAX_PLANS = 4     # a plan owns every activation of its shape (12 GB at batch 32 / 640 px): multi-scale training (train_one_epoch.py:66-70) walks
                      # through ~10 sizes, so only the most recently used plans are kept; tile choices live in the library, keyed by shape

    def plan(self, n, H, W, training):
        key = (n, H, W, bool(training), torch.cuda.current_stream().cuda_stream)
        p = self.plans.pop(key, None)
        if p is None:
            while len(self.plans) >= self.MAX_PLANS:
                torch.cuda.current_stream().synchronize()           # nothing of the evicted plan may still be running
                self.plans.pop(next(iter(self.plans)))
            # (tune.plan_build: MI355DET_TUNE_LOAD / _SAVE, and for N > 1 rank 0's timing choices broadcast to every rank.  The broadcast is a
            #  collective, so it runs only for plans every rank is known to build: training plans of an engine that takes part in data-parallel
            #  training - a GradSync is attached or SyncBN is on.  An evaluation loop, or a second engine that one rank builds for itself, may
            #  exist on a subset of the ranks)
            dp = bool(training and (self.sync_bn or getattr(self, "grad_syncs", ())))
            p = tune.plan_build(lambda: Plan(self, n, H, W, training, key[-1]), group=None, share=None if dp else False)
            if training:
                for gs in getattr(self, "grad_syncs", ()):       # parallel.GradSync.attach(): every plan gets the bucket hooks
                    gs.install(p)
        self.plans[key] = p                                          # most recently used last
        return p

    def forward(self, images, training=None):
        """images [n,3,H,W] fp32 NCHW on the GPU -> (out0,out1,out2) NCHW-shaped fp32 views [n,A*(5+C),h,w]."""
        training = self.training if training is None else training
        if images.dim() != 4 or images.shape[1] != 3 or not images.is_cuda:
            raise ValueError("expected a CUDA tensor [n,3,H,W]")
        n, _, H, W = images.shape
        if H % 32 or W % 32:
            raise ValueError("input size must be a multiple of 32")
        p = self.plan(n, H, W, training)
        p.run_forward(images.float().contiguous())
        self._last_plan = p
        return p.head_outputs()

    def train_step(self, images, targets, criterion, grad_scale=1.0):
        """One fused forward+backward (train_one_epoch.py:72-73,88-90 without the optimizer): the criterion's
        gradient goes straight into the bf16 head-gradient buffers, no autograd graph.  Returns out12
        (loss, sub_losses[6], stats[5]) as a device tensor."""
        heads = self.forward(images, training=True)
        p = self._last_plan
        p.zero_head_grads()
        gv, _keep = ops.head_views(p.head_grad_views(), self.head_c, dtype=self.adt)
        out12, _ = criterion._loss_impl(heads, targets, want_grad=True, grad_views=gv, grad_is_bf16=self.grad_fmt, grad_scale=grad_scale)
        p.run_backward()
        return out12

    def backward(self, head_grads=None):
        """Backward of the last training forward.  head_grads: None = the plan's bf16 head-gradient buffers were
        already filled (fused loss), or a list of NCHW fp32 tensors (autograd bridge)."""
        p = self._last_plan
        if head_grads is not None:
            p.load_head_grads(head_grads)
        p.run_backward()


class Plan:
    """Buffers + prepared call lists for one (batch, H, W, mode)."""

    def __init__(self, eng, n, H, W, training, stream):
        self.eng, self.n, self.H, self.W, self.training = eng, n, H, W, training
        self.stream = C.c_void_p(stream)
        self.fwd, self.bwd, self.pack = [], [], []
        self.fwd_const = []       # eval mode: folded BatchNorm rows (functions of the parameters only)
        self._const_epoch = -1
        self.keep = []            # ctypes structs / tensors that must outlive the call lists
        self.dz_elems = 0
        self.layers = {}
        dev = eng.device
        L = eng.L
        bf = eng.adt

        def new_act(n_, h_, w_, c_, buf=None, ld=None, off=0):
            if buf is None:
                buf = torch.zeros((n_, h_, w_, c_), device=dev, dtype=bf)
                ld = c_
            return Act(buf, n_, h_, w_, c_, ld, off)

        g5, g4, g3 = (H // 32, W // 32), (H // 16, W // 16), (H // 8, W // 8)
        self.cat1 = new_act(n, g4[0], g4[1], 768)
        self.cat2 = new_act(n, g3[0], g3[1], 384)
        self.heads = [torch.zeros((n, g[0], g[1], eng.head_ld), device=dev, dtype=torch.float32) for g in (g5, g4, g3)]
        self.head_grads = [torch.zeros((n, g[0], g[1], eng.head_ld), device=dev, dtype=bf) for g in (g5, g4, g3)]
        self.ops = []   # forward-ordered op records for the backward builder

        import torch.distributed as dist
        world = dist.get_world_size(eng.process_group) if (eng.sync_bn and training and dist.is_available() and dist.is_initialized()) else 1
        self.sync_world = world

        def sync_sum(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=eng.process_group)

        def sync_avg(t):
            # backward sums: the apply pass divides by the LOCAL element count, so the rank average gives the global mean (equal shards);
            # dgamma / dbeta then hold global / world on every rank, which the gradient all-reduce (an average) leaves unchanged - what
            # apex SyncBN + DDP produce
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=eng.process_group)
            t.div_(world)
        self._sync_avg = sync_avg

        # ---- forward graph
        def conv_bn(name, x, out=None, res=None, fused=None):
            """fused: (rows, emit(z, stats)) - a launch that produces this layer's z and partial statistics itself (stem + layer1.ds_conv)"""
            s = eng.by_name[name]
            shp = eng._wshape(s, x.n, x.h, x.w, in_ld=x.ld)
            x.conv_consumers += 1
            a = out if out is not None else new_act(x.n, shp.ho, shp.wo, shp.cout)
            wf, wd = eng.packed[name]
            b = bn_name(name)
            pixels = x.n * shp.ho * shp.wo
            cp = ops.cout_pad_of(shp.cout)
            ss = torch.zeros(4 * shp.cout, device=dev, dtype=torch.float32)
            if not training:
                # inference: BatchNorm uses the running statistics, so scale/shift are known BEFORE the convolution and the whole
                # BN + LeakyReLU (+ residual) runs in the MFMA epilogue; the pre-BN tensor z never exists (model.eval(),
                # test_one_epoch.py:10)
                shp.out_ld = a.ld
                e = _lib.ConvEpilogue(_vp(ss), _vp(ss, 4 * shp.cout), res.ptr if res else None, res.ld if res else 0, 2, 0, SLOPE)
                self.keep += [shp, ss, e]
                self.fwd_const.append((L.mi355det_bn_eval_scale_shift, (shp.cout, _vp(eng.params[b + ".weight"]), _vp(eng.params[b + ".bias"]),
                                                                  _vp(eng.buffers[b + ".running_mean"]),
                                                                  _vp(eng.buffers[b + ".running_var"]), BN_EPS, _vp(ss), self.stream)))
                if fused:
                    fused[1](a, ss)            # stem activation + this convolution + its folded BN / LeakyReLU in one launch
                else:
                    self.fwd.append((L.mi355det_conv_fwd_ex, (C.byref(shp), x.ptr, _vp(wf), C.byref(e), a.ptr, 0, cp, self.stream)))
                rec = dict(kind="cbl", name=name, spec=s, shp=shp, x=x, a=a, res=res, z=None, ss=ss, pixels=pixels)
                a.producer = rec
                self.ops.append(rec)
                self.layers[name] = rec
                return a
            shp.out_ld = shp.cout   # z pitch
            z = torch.zeros((x.n, shp.ho, shp.wo, shp.cout), device=dev, dtype=bf)
            rows = fused[0] if fused else L.mi355det_conv_stats_rows(C.byref(shp), ops.cout_pad_of(shp.cout))
            stats = torch.zeros((rows + 64, 2, cp), device=dev, dtype=torch.float32)
            self.keep += [shp, z, stats, ss]
            if fused:
                fused[1](z, stats)
            else:
                self.fwd.append((L.mi355det_conv_fwd, (C.byref(shp), x.ptr, _vp(wf), None, _vp(z), 0, _vp(stats), cp, self.stream)))
            if world > 1:
                # SyncBN: fold the partial rows into ONE row [sum | sum of squares] (the generic row reduction), all-reduce it, and
                # finalise from that row with the global element count
                row = torch.zeros((2, cp), device=dev, dtype=torch.float64)      # folded, all-reduced and finalised in double (the local path's precision)
                self.keep.append(row)
                self.fwd.append((L.mi355det_bn_fold_partials_f64, (_vp(stats), rows, shp.cout, cp, _vp(row), self.stream)))
                self.fwd.append((comm_hook, (sync_sum, row)))
                self.fwd.append((L.mi355det_bn_finalize_f64, (_vp(row), shp.cout, cp, pixels * world, _vp(eng.params[b + ".weight"]),
                                                              _vp(eng.params[b + ".bias"]), BN_EPS, BN_MOM,
                                                              _vp(eng.buffers[b + ".running_mean"]), _vp(eng.buffers[b + ".running_var"]),
                                                              _vp(ss), self.stream)))
            else:
                self.fwd.append((L.mi355det_bn_finalize, (_vp(stats), rows, shp.cout, cp, pixels, _vp(eng.params[b + ".weight"]),
                                                          _vp(eng.params[b + ".bias"]), BN_EPS, BN_MOM,
                                                          _vp(eng.buffers[b + ".running_mean"]), _vp(eng.buffers[b + ".running_var"]),
                                                          _vp(ss), self.stream)))
            self.fwd.append((L.mi355det_bn_act_fwd, (_vp(z), shp.cout, _vp(ss), shp.cout, pixels, SLOPE, res.ptr if res else None,
                                                     res.ld if res else 0, a.ptr, a.ld, self.stream)))
            self.dz_elems = max(self.dz_elems, pixels * shp.cout)
            rec = dict(kind="cbl", name=name, spec=s, shp=shp, x=x, a=a, res=res, z=z, ss=ss, pixels=pixels)
            a.producer = rec
            self.ops.append(rec)
            self.layers[name] = rec
            return a

        def conv_out(name, x, k):
            s = eng.by_name[name]
            x.conv_consumers += 1
            shp = eng._wshape(s, x.n, x.h, x.w, in_ld=x.ld, out_ld=eng.head_ld)
            shp_f = ops.conv_shape(x.n, x.h, x.w, s.cin, s.cout, 1, 1, x.ld, eng.head_ld)   # true cout: masks the pad channel
            wf, wd = eng.packed[name]
            cp = ops.cout_pad_of(shp_f.cout)
            self.keep += [shp, shp_f]
            self.fwd.append((L.mi355det_conv_fwd, (C.byref(shp_f), x.ptr, _vp(wf), _vp(eng.params[name + ".bias"]), _vp(self.heads[k]), 1,
                                                   None, cp, self.stream)))
            self.ops.append(dict(kind="out", name=name, spec=s, shp=shp, shp_f=shp_f, x=x, k=k))

        # stem (darknet.py:41-43,74-76): recompute kernels straight from the fp32 image, no stored z / im2col matrix (csrc/stem_kernels.hip)
        self.img_args = []       # argument lists whose first entry is the image pointer of the current step
        x = self._stem(n, H, W, new_act, sync_sum)
        stem_fused = self.layers["backbone.conv1"].get("fused_l1")
        feats = {}
        for li, nb in enumerate(BLOCKS[eng.backbone], 1):
            p = f"backbone.layer{li}"
            x = conv_bn(p + ".ds_conv", x, fused=stem_fused if li == 1 else None)
            for b in range(nb):
                y = conv_bn(f"{p}.residual_{b}.conv1", x)
                out = None
                if b == nb - 1 and li == 3:
                    out = self.cat2.slice(128, 256)      # torch.cat([x2_in, x2], 1): backbone slice after the upsampled one
                if b == nb - 1 and li == 4:
                    out = self.cat1.slice(256, 512)
                x = conv_bn(f"{p}.residual_{b}.conv2", y, out=out, res=x)
            feats[li] = x

        def branch(name, t, k):
            br = None
            for i in range(6):
                t = conv_bn(f"{name}.{i}.conv", t)
                if i == 4:
                    br = t
            conv_out(name + ".conv_out", t, k)
            return br
        b0 = branch("embedding0", feats[5], 0)
        t = conv_bn("embedding1_cbl.conv", b0)
        up1 = self.cat1.slice(0, 256)
        self.fwd.append((L.mi355det_upsample2x_fwd, (t.ptr, t.ld, t.n, t.h, t.w, t.c, up1.ptr, up1.ld, self.stream)))
        self.ops.append(dict(kind="up", x=t, cat=self.cat1, c_up=256, skip_to=feats[4]))
        b1 = branch("embedding1", self.cat1, 1)
        t = conv_bn("embedding2_cbl.conv", b1)
        up2 = self.cat2.slice(0, 128)
        self.fwd.append((L.mi355det_upsample2x_fwd, (t.ptr, t.ld, t.n, t.h, t.w, t.c, up2.ptr, up2.ld, self.stream)))
        self.ops.append(dict(kind="up", x=t, cat=self.cat2, c_up=128, skip_to=feats[3]))
        branch("embedding2", self.cat2, 2)

        # ---- weight packing (every step: the optimizer changes the fp32 masters) — one batched launch
        items = (_lib.PackItem * len(eng.specs))()
        for i, s in enumerate(eng.specs):
            shp = eng._wshape(s, 1, 8, 8)
            wf, wd = eng.packed[s.name]
            need_d = training and wd is not None
            items[i].w = eng.params[s.name + ".weight"].data_ptr()
            items[i].w_fwd = wf.data_ptr()
            items[i].w_dgrad = wd.data_ptr() if need_d else None
            items[i].shape = shp
            items[i].cout_pad = ops.cout_pad_of(shp.cout)
            items[i].w_is_ohwi = 1
        ne, nb = C.c_int32(0), C.c_int32(0)
        nbytes = L.mi355det_pack_table_bytes(items, len(eng.specs), C.byref(ne), C.byref(nb))
        host = torch.empty(nbytes, dtype=torch.uint8)
        check(L.mi355det_pack_table_build(items, len(eng.specs), C.c_void_p(host.data_ptr()), nbytes), "pack_table_build")
        self.pack_table = host.to(dev)
        self.pack.append((L.mi355det_pack_weights_batched, (_vp(self.pack_table), ne.value, nb.value, self.stream)))
        if training:
            self._build_backward()
            self._autotune()
        else:
            self._autotune_eval()

    def _stem(self, n, H, W, new_act, sync_sum):
        eng, L, dev = self.eng, self.eng.L, self.eng.device
        name, b = "backbone.conv1", "backbone.bn1"
        s = eng.by_name[name]
        wf, _ = eng.packed[name]
        a = new_act(n, H, W, 32)
        pixels = n * H * W
        rows = L.mi355det_stem_rows(n, H, W)
        if rows <= 0:
            raise ValueError("stem kernels need H % 8 == 0 and W % 32 == 0")
        ss = torch.zeros(4 * 32, device=dev, dtype=torch.float32)
        self.keep += [ss]

        def img_call(fn, args):
            args = [None] + list(args)
            self.img_args.append(args)
            return (fn, args)
        if not self.training:
            self.fwd_const.append((L.mi355det_bn_eval_scale_shift, (32, _vp(eng.params[b + ".weight"]), _vp(eng.params[b + ".bias"]),
                                                              _vp(eng.buffers[b + ".running_mean"]), _vp(eng.buffers[b + ".running_var"]),
                                                              BN_EPS, _vp(ss), self.stream)))
        else:
            part = torch.zeros((rows + 64, 2, 32), device=dev, dtype=torch.float32)
            self.keep.append(part)
            self.fwd.append(img_call(L.mi355det_stem_fwd_stats, (_vp(wf), _vp(part), n, H, W, self.stream)))
            fin = (_vp(eng.params[b + ".weight"]), _vp(eng.params[b + ".bias"]), BN_EPS, BN_MOM, _vp(eng.buffers[b + ".running_mean"]),
                   _vp(eng.buffers[b + ".running_var"]), _vp(ss), self.stream)
            if self.sync_world > 1:
                row = torch.zeros((2, 32), device=dev, dtype=torch.float64)
                self.keep.append(row)
                self.fwd.append((L.mi355det_bn_fold_partials_f64, (_vp(part), rows, 32, 32, _vp(row), self.stream)))
                self.fwd.append((comm_hook, (sync_sum, row)))
                self.fwd.append((L.mi355det_bn_finalize_f64, (_vp(row), 32, 32, pixels * self.sync_world) + fin))
            else:
                self.fwd.append((L.mi355det_bn_finalize, (_vp(part), rows, 32, 32, pixels) + fin))
        fused = None
        l1_rows = L.mi355det_stem_l1_rows(n, H, W)
        if self.training and l1_rows > 0:
            # training: the activation is produced INSIDE the kernel that convolves it (layer1.ds_conv, 32 -> 64 stride 2) and written to HBM
            # only as a side output for that layer's weight gradient (csrc/stem_l1_kernels.hip)
            wf1, _ = eng.packed["backbone.layer1.ds_conv"]

            def emit(z1, stats1):
                self.fwd.append(img_call(L.mi355det_stem_l1_fwd, (_vp(wf), _vp(ss), SLOPE, _vp(wf1), a.ptr, a.ld, _vp(z1), 64, _vp(stats1), n, H, W,
                                                                  self.stream)))
            fused = (l1_rows, emit)
        elif not self.training and l1_rows > 0:
            # inference: the same launch with layer 1's folded BN + LeakyReLU in its epilogue; the stem activation is never stored
            wf1, _ = eng.packed["backbone.layer1.ds_conv"]

            def emit_eval(a1, ss1):
                self.fwd.append(img_call(L.mi355det_stem_l1_fwd_eval, (_vp(wf), _vp(ss), SLOPE, _vp(wf1), _vp(ss1), a1.ptr, a1.ld, n, H, W, self.stream)))
            fused = (l1_rows, emit_eval)
        else:
            self.fwd.append(img_call(L.mi355det_stem_fwd_apply, (_vp(wf), _vp(ss), SLOPE, a.ptr, a.ld, n, H, W, self.stream)))
        rec = dict(kind="stem", name=name, spec=s, x=None, a=a, res=None, z=None, ss=ss, pixels=pixels, rows=rows, img_call=img_call, fused_l1=fused)
        a.producer = rec
        self.ops.append(rec)
        self.layers[name] = rec
        return a

    # ------------------------------------------------------------------
    def _build_backward(self):
        eng, L, dev = self.eng, self.eng.L, self.eng.device
        bf = eng.adt
        # two dz buffers (ping-pong) so the weight-gradient GEMM of layer L can run on a SECOND stream while the main
        # stream already does the BN backward / dgrad of the next layers: wgrad is off the dependency chain
        # (reduce -> apply -> dgrad), and the HBM-bound BN passes overlap with its MFMA work.
        self.dz2 = [torch.zeros(self.dz_elems, device=dev, dtype=bf) for _ in range(2)]
        self.dz = self.dz2[0]
        # BN-backward reduction fused into the producing dgrad's epilogue (mi355det_conv_dgrad_bn): correct and tested, but measured
        # SLOWER end to end in round 1 (849 vs 871 img/s): the z tile is read at the tile's end where nothing hides the HBM latency.
        # Opt-in until the prefetch is moved into the last k-steps.
        self.fuse_bn_reduce = eng.fuse_bn_reduce
        main = torch.cuda.current_stream(dev)
        # (one stream for everything was the A/B of round 3: +1.0 ms per step, profiles/r03_ab_results.md)
        self.side = torch.cuda.Stream(device=dev)
        side_ptr = C.c_void_p(self.side.cuda_stream)
        wg_done = [None, None]        # event: last wgrad that read dz2[i]
        flip = [0]

        def py(fn, *a):
            self.bwd.append((comm_hook, (fn,) + a))
        ws_need = max(L.mi355det_conv_wgrad_workspace(C.byref(r["shp_f"] if r["kind"] == "out" else r["shp"]))
                      for r in self.ops if r["kind"] in ("cbl", "out"))
        self.wgrad_ws = torch.empty(max(ws_need, 16), device=dev, dtype=torch.uint8)
        ws_ptr, ws_bytes = _vp(self.wgrad_ws), self.wgrad_ws.numel()
        nsum = sum(2 * r["shp"].cout for r in self.ops if r["kind"] == "cbl") + 64
        self.sums_all = torch.zeros(nsum, device=dev, dtype=torch.float32)
        sum_off = [0]
        # workspace of the fixed-order BatchNorm-backward sums (one partial row per workgroup, folded by a second small launch); one buffer
        # for every layer - the reduce launches are ordered on the step's stream
        red_need = max([L.mi355det_bn_act_bwd_reduce_workspace(r["shp"].cout, r["pixels"]) for r in self.ops if r["kind"] == "cbl"] + [1024])
        self.bn_red_ws = torch.zeros(red_need, device=dev, dtype=torch.uint8)
        red_ptr, red_bytes = _vp(self.bn_red_ws), self.bn_red_ws.numel()

        def grad_of(a):
            if a.grad is None:
                a.grad = Act(torch.zeros((a.n, a.h, a.w, a.c), device=dev, dtype=bf), a.n, a.h, a.w, a.c, a.c)
            return a.grad

        def emit_dgrad(shp, dy_ptr, wd, x):
            g = grad_of(x)
            if g.ld != shp.in_ld:     # x is a channel slice of a concat buffer; its gradient buffer is dense
                shp = _lib.ConvShape(shp.n, shp.h, shp.w, shp.cin, shp.ho, shp.wo, shp.cout, shp.ksize, shp.stride, shp.pad, g.ld,
                                     shp.out_ld)
                self.keep.append(shp)
            if not x.grad_written:
                r = x.skips.pop(0) if x.skips else None
                prod = getattr(x, "producer", None)
                # this dgrad writes the COMPLETE gradient of a BN+LeakyReLU activation (single conv consumer, at most one skip,
                # fused as the epilogue residual): start that layer's BatchNorm backward here, while the tile is on chip
                if prod is not None and prod.get("z") is not None and x.conv_consumers == 1 and not x.skips and self.fuse_bn_reduce:
                    rows = L.mi355det_conv_dgrad_bn_rows(C.byref(shp))
                    cpad = ops.pad_to(x.c, 32)
                    part = torch.zeros((rows + 64, 2, cpad), device=dev, dtype=torch.float32)
                    prod["bn_partials"] = (part, rows, cpad)
                    self.bwd.append((L.mi355det_conv_dgrad_bn, (C.byref(shp), dy_ptr, _vp(wd), g.ptr, r.ptr if r else None, r.ld if r else 0,
                                                                _vp(prod["z"]), x.c, _vp(prod["ss"]), SLOPE, _vp(part), self.stream)))
                else:
                    self.bwd.append((L.mi355det_conv_dgrad, (C.byref(shp), dy_ptr, _vp(wd), g.ptr, r.ptr if r else None, r.ld if r else 0,
                                                             self.stream)))
                x.grad_written = True
            else:
                self.bwd.append((L.mi355det_conv_dgrad, (C.byref(shp), dy_ptr, _vp(wd), g.ptr, g.ptr, g.ld, self.stream)))
            while x.skips:
                r = x.skips.pop(0)
                self.bwd.append((L.mi355det_add_bf16, (g.ptr, g.ld, r.ptr, r.ld, x.c, x.pixels, g.ptr, g.ld, self.stream)))

        self.bwd_marks = []      # (index into self.bwd after the layer's calls, lowest flat_g offset completed)
        first_off = {name: o for name, o, _n, _s in eng.param_order}
        for rec in reversed(self.ops):
            if rec["kind"] in ("out", "cbl", "stem"):
                self._mark_name = rec["name"]
            if rec["kind"] == "stem":
                # BatchNorm backward + weight gradient of the stem from the image and the activation gradient alone: z and dz are
                # recomputed in registers (two passes: the per-channel sums, then dz straight into the weight-gradient MFMA)
                name, a, ss, rows, img_call = rec["name"], rec["a"], rec["ss"], rec["rows"], rec["img_call"]
                assert a.grad is not None and a.grad_written and not a.skips, name
                g, b = a.grad, bn_name(name)
                wf, _ = eng.packed[name]
                sums = self.sums_all[sum_off[0]:sum_off[0] + 64]
                sum_off[0] += 64
                # (the round-3 two-pass form - BN sums, then dz into the weight-gradient MFMA - lost its A/B by 0.2-0.4 ms per step and is gone from the
                #  engine; its entry points mi355det_stem_bwd_reduce / _apply_wgrad stay as the references of tests/test_gpu_stem.py)
                # ONE pass: A = dy^T [im2col | 1] and the Gram matrix of [im2col | 1] on MFMA; the BN sums and dW are linear in them
                slab = torch.zeros((rows, 2048), device=dev, dtype=torch.float32)
                ag = torch.zeros(2048, device=dev, dtype=torch.float32)
                self.keep += [slab, ag]
                self.bwd.append(img_call(L.mi355det_stem_bwd_fused, (_vp(wf), _vp(ss), SLOPE, g.ptr, g.ld, _vp(slab), _vp(ag), _vp(sums),
                                                                     self.n, self.H, self.W, self.stream)))
                if self.sync_world > 1:
                    py(self._sync_avg, sums)
                self.bwd.append((L.mi355det_stem_bwd_finish, (_vp(wf), _vp(ss), _vp(ag), _vp(sums), self.n * self.H * self.W,
                                                              _vp(eng.grads[name + ".weight"]), _vp(eng.grads[b + ".weight"]),
                                                              _vp(eng.grads[b + ".bias"]), self.stream)))
                ev_stem = torch.cuda.Event()
                py(ev_stem.record, main)                  # the stem's gradients come from the main stream: the side stream (last
                py(self.side.wait_event, ev_stem)         # gradient bucket) must see them
            elif rec["kind"] == "out":
                shp, shp_f, x, k, name = rec["shp"], rec["shp_f"], rec["x"], rec["k"], rec["name"]
                _, wd = eng.packed[name]
                dy = _vp(self.head_grads[k])
                ev = torch.cuda.Event()
                py(ev.record, main)                       # head gradients ready (criterion ran on the main stream)
                py(self.side.wait_event, ev)
                self.bwd.append((L.mi355det_conv_wgrad, (C.byref(shp_f), x.ptr, dy, _vp(eng.grads[name + ".weight"]),
                                                         _vp(eng.grads[name + ".bias"]), ws_ptr, ws_bytes, side_ptr)))
                emit_dgrad(shp, dy, wd, x)
            elif rec["kind"] == "up":
                x, cat, c_up, skip_to = rec["x"], rec["cat"], rec["c_up"], rec["skip_to"]
                assert cat.grad is not None and cat.grad_written
                gup = cat.grad.slice(0, c_up)
                gx = grad_of(x)
                self.bwd.append((L.mi355det_upsample2x_bwd, (gup.ptr, gup.ld, x.n, x.h, x.w, x.c, gx.ptr, gx.ld, self.stream)))
                x.grad_written = True
                skip_to.skips.append(cat.grad.slice(c_up, cat.c - c_up))
            else:
                name, shp, x, a, res, z, ss, pixels = (rec[k] for k in ("name", "shp", "x", "a", "res", "z", "ss", "pixels"))
                assert a.grad is not None and a.grad_written and not a.skips, name
                g = a.grad
                b = bn_name(name)
                sums = self.sums_all[sum_off[0]:sum_off[0] + 2 * shp.cout]
                sum_off[0] += 2 * shp.cout
                if res is not None:
                    res.skips.append(g)
                di = flip[0]
                flip[0] ^= 1
                dzb = self.dz2[di]
                rec["dz_index"] = di
                if "bn_partials" in rec:      # the dgrad that produced g already accumulated the per-channel partial sums
                    part, prows, cpad = rec["bn_partials"]
                    self.bwd.append((L.mi355det_bn_bwd_sum_partials, (_vp(part), prows, shp.cout, cpad, _vp(sums), self.stream)))
                elif eng.deterministic:
                    self.bwd.append((L.mi355det_bn_act_bwd_reduce_det, (g.ptr, g.ld, None, 0, _vp(z), shp.cout, _vp(ss), shp.cout, pixels, SLOPE,
                                                                        _vp(sums), red_ptr, red_bytes, self.stream)))
                else:
                    self.bwd.append((L.mi355det_bn_act_bwd_reduce, (g.ptr, g.ld, None, 0, _vp(z), shp.cout, _vp(ss), shp.cout, pixels, SLOPE,
                                                                    _vp(sums), self.stream)))
                if self.sync_world > 1:
                    py(self._sync_avg, sums)              # SyncBN backward: (sum dy, sum dy*xhat) of the global batch
                if wg_done[di] is not None:
                    py(main.wait_event, wg_done[di])      # the wgrad that last read this dz buffer has finished
                self.bwd.append((L.mi355det_bn_act_bwd_apply, (g.ptr, g.ld, None, 0, _vp(z), shp.cout, _vp(ss), _vp(sums), None, shp.cout,
                                                               pixels, SLOPE, _vp(dzb), shp.cout, _vp(eng.grads[b + ".weight"]),
                                                               _vp(eng.grads[b + ".bias"]), self.stream)))
                # (measured and removed: the weight gradients of the large maps serial on the main stream - no gain, profiles/r03_ab_results.md)
                # (measured and removed, round 4: the weight gradient started only when its layer's data gradient has finished, so that it runs beside
                #  the next BatchNorm passes instead of beside the data gradient: 28.17 -> 29.85 ms, profiles/r04_ab_results.md 8 - the two GEMMs
                #  side by side fill each other's tails; serialised they do not)
                ev_dz, ev_wg = torch.cuda.Event(), torch.cuda.Event()
                py(ev_dz.record, main)
                py(self.side.wait_event, ev_dz)
                self.bwd.append((L.mi355det_conv_wgrad, (C.byref(shp), x.ptr, _vp(dzb), _vp(eng.grads[name + ".weight"]), None,
                                                         ws_ptr, ws_bytes, side_ptr)))
                py(ev_wg.record, self.side)
                wg_done[di] = ev_wg
                _, wd = eng.packed[name]
                emit_dgrad(shp, _vp(dzb), wd, x)
            if rec["kind"] in ("out", "cbl", "stem"):
                self.bwd_marks.append((len(self.bwd), first_off[rec["name"] + ".weight"]))
        ev_end = torch.cuda.Event()
        py(ev_end.record, self.side)
        py(main.wait_event, ev_end)                       # join: backward is complete on the main stream
        self.side_stream = self.side

    def _autotune(self):
        """Plan-build time only: let the library time its candidate tile configurations / split counts for every
        conv shape of this plan on the plan's own buffers (mi355det_conv_autotune_mode, _wgrad_autotune)."""
        eng, L = self.eng, self.eng.L
        saved = {k: v.clone() for k, v in eng.buffers.items()}
        img = torch.randn((self.n, 3, self.H, self.W), device=eng.device)
        self._set_image(img)
        for g in self.head_grads:
            g.normal_(0, 1e-2)
        L.mi355det_conv_autotune_mode(1)
        try:
            self._run(self.pack)
            self._run(self.fwd)
            self._run(self.bwd)
        finally:
            L.mi355det_conv_autotune_mode(0)
        ws_ptr, ws_bytes = _vp(self.wgrad_ws), self.wgrad_ws.numel()
        for rec in self.ops:
            if rec["kind"] == "cbl":
                st = L.mi355det_conv_wgrad_autotune(C.byref(rec["shp"]), rec["x"].ptr, _vp(self.dz), _vp(eng.grads[rec["name"] + ".weight"]),
                                                    ws_ptr, ws_bytes, self.stream)
            elif rec["kind"] == "out":
                st = L.mi355det_conv_wgrad_autotune(C.byref(rec["shp_f"]), rec["x"].ptr, _vp(self.head_grads[rec["k"]]),
                                                    _vp(eng.grads[rec["name"] + ".weight"]), ws_ptr, ws_bytes, self.stream)
            else:
                continue
            if st < 0:
                check(st, "conv_wgrad_autotune")
        torch.cuda.synchronize()
        for k, v in saved.items():
            eng.buffers[k].copy_(v)
        eng.flat_g.zero_()
        self.zero_head_grads()

    def _autotune_eval(self):
        """Eval plans run the convolutions with the BatchNorm + LeakyReLU (+ residual) epilogue, whose tile choices are keyed separately
        from the training forward's: time the candidates once on this plan's buffers (round 2 ran every eval convolution on the default
        128 x 128 tile: 9.2 ms against 7.0 ms for the same 75 convolutions in the training step)."""
        L = self.eng.L
        img = torch.randn((self.n, 3, self.H, self.W), device=self.eng.device)
        self._set_image(img)
        self._run(self.pack)
        self._run(self.fwd_const)
        L.mi355det_conv_autotune_mode(1)
        try:
            self._run(self.fwd)
        finally:
            L.mi355det_conv_autotune_mode(0)
        torch.cuda.synchronize()

    # ------------------------------------------------------------------
    def _run(self, calls):
        for fn, args in calls:
            if fn is comm_hook:
                args[0](*args[1:])
                continue
            st = fn(*args)
            if st != 0:
                check(st, fn.__name__)

    def _set_image(self, images):
        """The stem kernels of BOTH directions read the fp32 image: it stays referenced until the next forward."""
        self._img = images
        ptr = C.c_void_p(images.data_ptr())
        for a in self.img_args:
            a[0] = ptr

    def run_forward(self, images):
        self._set_image(images)
        eng = self.eng
        if self.training:
            eng._weights_changed()                 # an optimizer step may follow
            self._run(self.pack)
        elif not (eng._frozen and self._const_epoch == eng._static_epoch):
            self._run(self.pack)
            self._run(self.fwd_const)
            self._const_epoch = eng._static_epoch if eng._frozen else -1
        self._run(self.fwd)
        if self.training:
            eng.num_batches_tracked += 1

    def head_outputs(self):
        c = self.eng.head_c
        return [h[..., :c].permute(0, 3, 1, 2) for h in self.heads]

    def head_grad_views(self):
        """bf16 gradient buffers behind the same NCHW-shaped views (for the fused criterion)."""
        c = self.eng.head_c
        return [g[..., :c].permute(0, 3, 1, 2) for g in self.head_grads]

    def zero_head_grads(self):
        for g in self.head_grads:
            g.zero_()

    def load_head_grads(self, grads):
        c = self.eng.head_c
        for dst, g in zip(self.head_grads, grads):
            dst[..., :c].copy_(g.permute(0, 2, 3, 1))

    def run_backward(self):
        self.eng.flat_g.zero_()
        self.sums_all.zero_()
        self.side.wait_stream(torch.cuda.current_stream())   # zeroed gradients / forward activations visible to the side stream
        self._run(self.bwd)


def comm_hook(*a):   # marker: (comm_hook, (callable, *args)) entries run a python callback inside a call list
    raise RuntimeError("marker only")
