"""RetinaNet-ResNet50-FPN executor over libmi355det.so (BASELINE config 3).

Mirrors the forward graph of `retinanet_resnet50_fpn` (tvision/retinanet.py:583-660): ResNet-50 body with
FrozenBatchNorm2d (utilities/resnet.py:87-143,230-240; backbone_utils.py:67-110), FeaturePyramidNetwork + LastLevelP6P7
(backbone_utils.py:33-63), RetinaNetHead (retinanet.py:40-246), and owns the matching backward.  MI355X-first choices:

  * FrozenBatchNorm2d never exists as a pass: its per-channel scale/shift (+ ReLU, + the Bottleneck identity add) run in
    the convolution's MFMA epilogue, so a bottleneck is 3-4 launches forward; backward needs one mask*scale pass per conv;
  * the heads write fp32 logits straight into the level-concatenated [N, sum HWA, K] tensors the reference builds with
    permute+reshape+cat (retinanet.py:163-170), and the whole batch's loss + gradient is three launches;
  * NHWC bf16 activations resident for the step, fp32 master weights (OHWI) in ONE flat buffer for the trainable part
    (forward order -> DDP buckets are tail slices), frozen weights packed once;
  * a step is a static list of prepared C-ABI calls; weight-gradient GEMMs run on a second stream.
"""
import ctypes as C
import math
import os

import torch

from .. import _lib, ops, tune
from .._lib import check, lib
from ..yolo.nets.engine import Act, comm_hook, _vp
from .anchor_utils import AnchorGenerator

LAYERS = [3, 4, 6, 3]
BODY_LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3], "resnet152": [3, 8, 36, 3]}     # utilities/resnet.py:305-341
PLANES = [64, 128, 256, 512]
STEM_K = 160        # 7*7*3 = 147 im2col columns padded to a multiple of 32
IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
ANCHOR_SIZES = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in [32, 64, 128, 256, 512])   # retinanet.py:647
ASPECT_RATIOS = ((0.5, 1.0, 2.0),) * 5


class Conv:
    def __init__(self, name, cin, cout, k, stride, bn=None, bias=False, relu=False, trainable=True, head=None):
        self.name, self.cin, self.cout, self.k, self.stride = name, cin, cout, k, stride
        self.bn, self.bias, self.relu, self.trainable, self.head = bn, bias, relu, trainable, head
        # the data gradient reduces over cout: a multiple of 64 lets it use the 64-deep k-step kernels (igemm8, shared pixel tiles) - with 32 the
        # K = 1204 cls_logits layer (9 * 1204 = 10836 channels) was left to the 128x128x32 tile: 17 ms of a 55 ms step at 314 TFLOP/s
        self.cout_store = ops.pad_to(cout, 64) if head else cout


def arch(num_classes=91, num_anchors=9, trainable_layers=3, body="resnet50", model="retinanet"):
    """Ordered conv specs (reference state_dict order).  Frozen: everything in the body below the last
    `trainable_layers` of [layer4, layer3, layer2, layer1, conv1] (backbone_utils.py:100-104); BN is always frozen."""
    if not 0 <= trainable_layers <= 5:
        raise ValueError("trainable_backbone_layers must be in [0,5]")          # backbone_utils.py:100 `assert 0 <= trainable_layers <= 5`
    train = set(["layer4", "layer3", "layer2", "layer1", "conv1"][:trainable_layers])
    B = "backbone.body."
    # 5 also lists 'bn1' (backbone_utils.py:103-104), which is a FrozenBatchNorm2d: buffers only, nothing becomes trainable
    specs = [Conv(B + "conv1", STEM_K, 64, 1, 1, bn=B + "bn1", relu=True, trainable="conv1" in train)]
    inpl = 64
    for li, (planes, nb) in enumerate(zip(PLANES, BODY_LAYERS[body]), 1):
        tr = f"layer{li}" in train
        for b in range(nb):
            q = f"{B}layer{li}.{b}"
            s = 2 if (b == 0 and li > 1) else 1
            specs.append(Conv(q + ".conv1", inpl, planes, 1, 1, bn=q + ".bn1", relu=True, trainable=tr))
            specs.append(Conv(q + ".conv2", planes, planes, 3, s, bn=q + ".bn2", relu=True, trainable=tr))
            specs.append(Conv(q + ".conv3", planes, planes * 4, 1, 1, bn=q + ".bn3", relu=True, trainable=tr))
            if b == 0:
                specs.append(Conv(q + ".downsample.0", inpl, planes * 4, 1, s, bn=q + ".downsample.1", trainable=tr))
            inpl = planes * 4
    Fp = "backbone.fpn."
    if model == "fasterrcnn":
        # resnet_fpn_backbone(returned_layers=[1,2,3,4], extra_blocks=LastLevelMaxPool) (backbone_utils.py:106-122) + RPNHead (rpn.py:17-58)
        for i, cin in enumerate((256, 512, 1024, 2048)):
            specs.append(Conv(f"{Fp}inner_blocks.{i}", cin, 256, 1, 1, bias=True))
        for i in range(4):
            specs.append(Conv(f"{Fp}layer_blocks.{i}", 256, 256, 3, 1, bias=True))
        specs.append(Conv("rpn.head.conv", 256, 256, 3, 1, bias=True, relu=True))
        specs.append(Conv("rpn.head.cls_logits", 256, num_anchors * num_classes, 1, 1, bias=True, head="cls_logits"))
        specs.append(Conv("rpn.head.bbox_pred", 256, num_anchors * 4, 1, 1, bias=True, head="bbox_reg"))
        return specs
    for i, cin in enumerate((512, 1024, 2048)):
        specs.append(Conv(f"{Fp}inner_blocks.{i}", cin, 256, 1, 1, bias=True))
    for i in range(3):
        specs.append(Conv(f"{Fp}layer_blocks.{i}", 256, 256, 3, 1, bias=True))
    specs.append(Conv(Fp + "extra_blocks.p6", 256, 256, 3, 2, bias=True))
    specs.append(Conv(Fp + "extra_blocks.p7", 256, 256, 3, 2, bias=True))
    for hname, last, cout in (("classification_head", "cls_logits", num_anchors * num_classes), ("regression_head", "bbox_reg", num_anchors * 4)):
        for i in (0, 2, 4, 6):
            specs.append(Conv(f"head.{hname}.conv.{i}", 256, 256, 3, 1, bias=True, relu=True))
        specs.append(Conv(f"head.{hname}.{last}", 256, cout, 3, 1, bias=True, head=last))
    return specs


class RetinaNetEngine:
    MODEL = "retinanet"

    def __init__(self, num_classes=91, num_anchors=9, trainable_layers=3, device=None, seed=0, bn_eps=1e-5, normalize=True, body="resnet50"):
        lib()   # fail loudly if the HIP library is missing
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.nc, self.na = num_classes, num_anchors
        self.bn_eps = bn_eps
        self.normalize = normalize
        self.body_name = body
        self.model = self.MODEL
        self.specs = arch(num_classes, num_anchors, trainable_layers, body, self.MODEL)
        self.by_name = {s.name: s for s in self.specs}
        self._layout_params()
        self.reset_parameters(seed)
        self.plans = {}
        self.training = True
        if self.model == "fasterrcnn":
            self.anchor_generator = AnchorGenerator(((32,), (64,), (128,), (256,), (512,)), ((0.5, 1.0, 2.0),) * 5)     # frcnn.py:187-190
        else:
            self.anchor_generator = AnchorGenerator(ANCHOR_SIZES, ASPECT_RATIOS)
        self.mean = torch.tensor(IMAGE_MEAN, device=self.device)
        self.inv_std = 1.0 / torch.tensor(IMAGE_STD, device=self.device)

    # ------------------------------------------------------------------ parameters
    def _layout_params(self):
        dev = self.device
        offs = {True: 0, False: 0}
        order = {True: [], False: []}
        for s in self.specs:
            shape = (s.cout_store, s.k, s.k, s.cin)
            n = math.prod(shape)
            order[s.trainable].append((s.name + ".weight", offs[s.trainable], n, shape))
            offs[s.trainable] += ops.pad_to(n, 64)
            if s.bias:
                order[s.trainable].append((s.name + ".bias", offs[s.trainable], s.cout_store, (s.cout_store,)))
                offs[s.trainable] += ops.pad_to(s.cout_store, 64)
        self.flat_w = torch.zeros(max(offs[True], 64), device=dev)
        self.flat_g = torch.zeros(max(offs[True], 64), device=dev)
        self.frozen_w = torch.zeros(max(offs[False], 64), device=dev)
        self.param_order = order[True]
        self.params, self.grads = {}, {}
        for name, o, n, shape in order[True]:
            self.params[name] = self.flat_w[o:o + n].view(shape)
            self.grads[name] = self.flat_g[o:o + n].view(shape)
        for name, o, n, shape in order[False]:
            self.params[name] = self.frozen_w[o:o + n].view(shape)
        # FrozenBatchNorm2d buffers and the folded (scale, shift) they reduce to
        self.buffers, self.affine = {}, {}
        for s in self.specs:
            if s.bn:
                for k, v in ((".weight", 1.0), (".bias", 0.0), (".running_mean", 0.0), (".running_var", 1.0)):
                    self.buffers[s.bn + k] = torch.full((s.cout,), v, device=dev)
                self.affine[s.bn] = torch.zeros((2, s.cout), device=dev)
        self.packed = {}
        for s in self.specs:
            shp = self._shape(s, 1, 8, 8)
            cp = ops.cout_pad_of(s.cout)
            wf = torch.zeros(cp * s.k * s.k * s.cin, device=dev, dtype=torch.bfloat16)
            wd = torch.zeros(lib().mi355det_dgrad_pack_elems(C.byref(shp)), device=dev, dtype=torch.bfloat16) if s.name != "backbone.body.conv1" else None
            self.packed[s.name] = (wf, wd)

    def _shape(self, s, n, h, w, store=True, in_ld=None, out_ld=None):
        return ops.conv_shape(n, h, w, s.cin, s.cout_store if store else s.cout, s.k, s.stride, in_ld, out_ld)

    def reset_parameters(self, seed=0):
        """Reference initialisation: kaiming-normal(fan_out) body convs (resnet.py:195-197), kaiming-uniform(a=1) FPN with zero
        bias (torchvision FPN), N(0, 0.01) heads with the prior-probability cls bias (retinanet.py:86-97,203-211)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        for s in self.specs:
            kk, cin = s.k * s.k, (3 if s.name.endswith("body.conv1") else s.cin)
            ks = 7 if s.name.endswith("body.conv1") else s.k
            if s.name.startswith("backbone.body."):
                t = torch.randn((s.cout, cin, ks, ks), generator=g) * math.sqrt(2.0 / (s.cout * ks * ks))
            elif s.name.startswith("backbone.fpn."):
                bound = math.sqrt(6.0 / ((1 + 1.0) * cin * kk))
                t = (torch.rand((s.cout, cin, ks, ks), generator=g) * 2 - 1) * bound
            else:
                t = torch.randn((s.cout, cin, ks, ks), generator=g) * 0.01
            self._set_weight_oihw(s, t)
            if s.bias:
                self.params[s.name + ".bias"].zero_()
                if s.name == "head.classification_head.cls_logits":
                    self.params[s.name + ".bias"][:s.cout].fill_(-math.log((1 - 0.01) / 0.01))
        self.refresh_frozen()

    def _set_weight_oihw(self, s, t):
        w = self.params[s.name + ".weight"]
        t = t.to(self.device, torch.float32)
        w.zero_()
        if s.name.endswith("body.conv1"):
            w.view(64, STEM_K)[:, :147] = t.permute(0, 2, 3, 1).reshape(64, 147)        # k = (kh*7+kw)*3 + c
        else:
            w[:s.cout] = t.permute(0, 2, 3, 1)

    def _get_weight_oihw(self, s, src):
        w = src[s.name + ".weight"]
        if s.name.endswith("body.conv1"):
            return w.view(64, STEM_K)[:, :147].reshape(64, 7, 7, 3).permute(0, 3, 1, 2).contiguous()
        return w[:s.cout].permute(0, 3, 1, 2).contiguous()

    def load_reference_state_dict(self, sd):
        """state_dict of the reference `RetinaNet` (keys as in retinanet_resnet50_fpn(...).state_dict())."""
        for s in self.specs:
            self._set_weight_oihw(s, sd[s.name + ".weight"])
            if s.bias:
                self.params[s.name + ".bias"].zero_()
                self.params[s.name + ".bias"][:s.cout].copy_(sd[s.name + ".bias"])
            if s.bn:
                for k in (".weight", ".bias", ".running_mean", ".running_var"):
                    self.buffers[s.bn + k].copy_(sd[s.bn + k])
        self.refresh_frozen()

    def reference_state_dict(self, grads=False):
        out = {}
        for s in self.specs:
            if grads and not s.trainable:
                continue
            src = self.grads if grads else self.params
            out[s.name + ".weight"] = self._get_weight_oihw(s, src)
            if s.bias:
                out[s.name + ".bias"] = src[s.name + ".bias"][:s.cout].clone()
            if s.bn and not grads:
                for k in (".weight", ".bias", ".running_mean", ".running_var"):
                    out[s.bn + k] = self.buffers[s.bn + k].clone()
        return out

    def refresh_frozen(self):
        """Fold FrozenBatchNorm2d into per-channel (scale, shift) and (re)pack the frozen convolution weights — after loading."""
        for s in self.specs:
            if s.bn:
                w, b, rm, rv = (self.buffers[s.bn + k] for k in (".weight", ".bias", ".running_mean", ".running_var"))
                scale = w * torch.rsqrt(rv + self.bn_eps)
                self.affine[s.bn][0].copy_(scale)
                self.affine[s.bn][1].copy_(b - rm * scale)
        self._pack([s for s in self.specs if not s.trainable], need_dgrad=False)

    def _pack_table(self, specs, need_dgrad):
        L = lib()
        items = (_lib.PackItem * max(1, len(specs)))()
        for i, s in enumerate(specs):
            wf, wd = self.packed[s.name]
            items[i].w = self.params[s.name + ".weight"].data_ptr()
            items[i].w_fwd = wf.data_ptr()
            items[i].w_dgrad = wd.data_ptr() if (need_dgrad and wd is not None) else None
            items[i].shape = self._shape(s, 1, 8, 8)
            items[i].cout_pad = ops.cout_pad_of(s.cout)
            items[i].w_is_ohwi = 1
        ne, nb = C.c_int32(0), C.c_int32(0)
        nbytes = L.mi355det_pack_table_bytes(items, len(specs), C.byref(ne), C.byref(nb))
        host = torch.empty(max(nbytes, 1), dtype=torch.uint8)
        check(L.mi355det_pack_table_build(items, len(specs), C.c_void_p(host.data_ptr()), nbytes), "pack_table_build")
        return host.to(self.device), ne.value, nb.value

    def _pack(self, specs, need_dgrad):
        if not specs:
            return
        tab, ne, nb = self._pack_table(specs, need_dgrad)
        check(lib().mi355det_pack_weights_batched(_vp(tab), ne, nb, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pack_weights_batched")
        torch.cuda.current_stream().synchronize()      # `tab` is freed on return

    # ------------------------------------------------------------------ plan / step
    MAX_PLANS = 4     # a plan owns every activation buffer of its shape: keep the few most recently used (multi-size inputs, train + eval)

    def plan(self, n, H, W, training):
        key = (n, H, W, bool(training), bool(self.normalize), torch.cuda.current_stream().cuda_stream)
        p = self.plans.pop(key, None)
        if p is None:
            while len(self.plans) >= self.MAX_PLANS:
                torch.cuda.current_stream().synchronize()           # nothing of the evicted plan may still be running
                self.plans.pop(next(iter(self.plans)))
            # (tune.plan_build: MI355DET_TUNE_LOAD / _SAVE; for N > 1 rank 0's timing choices are broadcast - a collective, so only for training
            #  plans of an engine with a GradSync attached: plans every rank is known to build)
            dp = bool(training and getattr(self, "grad_syncs", ()))
            p = tune.plan_build(lambda: RetinaPlan(self, n, H, W, training, key[-1]), share=None if dp else False)
            if training:
                for gs in getattr(self, "grad_syncs", ()):       # parallel.GradSync.attach(): every plan gets the bucket hooks
                    gs.install(p)
        self.plans[key] = p                                          # most recently used last
        return p

    def forward(self, images, training=None):
        """images [n,3,H,W] fp32 in 0..1 (already resized/batched) -> {'cls_logits': [n, sum HWA, K], 'bbox_regression': [n, sum HWA, 4]}."""
        training = self.training if training is None else training
        if images.dim() != 4 or images.shape[1] != 3 or not images.is_cuda:
            raise ValueError("expected a CUDA tensor [n,3,H,W]")
        n, _, H, W = images.shape
        if H % 32 or W % 32:
            raise ValueError("input size must be a multiple of 32 (GeneralizedRCNNTransform.batch_images size_divisible)")
        p = self.plan(n, H, W, training)
        p.run_forward(images.float().contiguous())
        self._last_plan = p
        return {"cls_logits": p.logits, "bbox_regression": p.bbox_reg}

    def match(self, p, targets):
        """RetinaNet.compute_loss matching (retinanet.py:401-412): fused IoU + Matcher(0.5, 0.4, allow_low_quality) per image."""
        for i, t in enumerate(targets):
            if t["boxes"].numel() == 0:
                p.matched[i].fill_(-1)
            else:
                p.matched[i].copy_(ops.match_anchors(t["boxes"], p.anchors, 0.5, 0.4, True))
        return p.matched

    def train_step(self, images, targets, class_scale=None, grad_scale=1.0):
        """Forward + RetinaNetHead.compute_loss + backward.  targets: list of {'boxes' [M,4] xyxy, 'labels' [M] int64}.
        Returns losses[2] = (classification, bbox_regression) as a device tensor."""
        # Everything that depends only on the ground truth is issued BEFORE the network forward: the matching kernels run first on an idle
        # device, and the host-to-device copy of the offsets (stream-ordered: built after the forward it made the host wait for the whole
        # forward, 8 ms at batch 16, and issue the loss and the backward behind it with the device idle for ~1 ms) costs nothing there.
        if images.dim() != 4 or images.shape[1] != 3 or not images.is_cuda:
            raise ValueError("expected a CUDA tensor [n,3,H,W]")
        if images.shape[2] % 32 or images.shape[3] % 32:
            raise ValueError("input size must be a multiple of 32 (GeneralizedRCNNTransform.batch_images size_divisible)")
        p = self.plan(images.shape[0], images.shape[2], images.shape[3], True)
        counts = [int(t["boxes"].shape[0]) for t in targets]
        offs_host = torch.tensor([0] + [sum(counts[:i + 1]) for i in range(len(counts))], dtype=torch.int32).pin_memory()
        offs = offs_host.to(self.device, non_blocking=True)
        if sum(counts):
            gt_boxes = torch.cat([t["boxes"].reshape(-1, 4).float() for t in targets])
            gt_labels = torch.cat([t["labels"].reshape(-1).long() for t in targets])
        else:
            gt_boxes, gt_labels = torch.zeros((1, 4), device=self.device), torch.zeros(1, dtype=torch.int64, device=self.device)
        self.match(p, targets)
        self.forward(images, training=True)
        assert self._last_plan is p
        # the class gradient goes straight into the bf16 per-level buffers the cls_logits backward reads (no fp32 gradient tensor, no cast:
        # the round-2 route through fp32 glogits + cast_rows was 2.0 ms slower per R101-LVIS step, profiles/r03_ab_results.md)
        levels = [p.head_grads[("cls_logits", lvl)] for lvl in range(len(p.level_sizes))]
        losses, nfg, _, _ = ops.retina_loss(p.logits, p.bbox_reg, p.anchors, p.matched, gt_boxes, gt_labels, offs, class_scale=class_scale,
                                            grad_scale=grad_scale, grad_regression=p.gbbox, cls_levels=levels, anchors_per_pixel=self.na)
        p.load_head_grads(cls=False)
        self.last_num_foreground = nfg
        p.run_backward()
        return losses

    def backward(self, grad_logits, grad_regression):
        """Backward of the last training forward for externally supplied head gradients (autograd bridge / tests)."""
        p = self._last_plan
        p.glogits.copy_(grad_logits)
        p.gbbox.copy_(grad_regression)
        p.load_head_grads()
        p.run_backward()


class FasterRCNNEngine(RetinaNetEngine):
    """ResNet-FPN backbone (C2..C5 -> P2..P5 + max-pool level) + RPNHead of `fasterrcnn_resnet50_fpn` (tvision/frcnn.py:150-236,
    backbone_utils.py:67-122, rpn.py:17-58).  The RPN outputs are the level-concatenated objectness [N, sum HWA, 1] and deltas
    [N, sum HWA, 4] (the layout rpn.py:concat_box_prediction_layers builds); P2..P5 are handed to the RoI heads as NCHW fp32
    tensors and their gradients come back through `backward(..., feature_grads)`."""
    MODEL = "fasterrcnn"

    def __init__(self, trainable_layers=3, device=None, seed=0, bn_eps=1e-5, normalize=True, body="resnet50"):
        super().__init__(num_classes=1, num_anchors=3, trainable_layers=trainable_layers, device=device, seed=seed, bn_eps=bn_eps,
                         normalize=normalize, body=body)

    def train_step(self, *a, **k):
        raise NotImplementedError("use tvision.frcnn.FasterRCNN: the step has a data-dependent middle (proposals, sampling)")

    def feature_maps_nchw(self, levels=4):
        """P2.. as NCHW fp32 tensors (the layout mi355det_roi_align reads)."""
        p = self._last_plan
        outs = []
        for f in p.features[:levels]:
            o = torch.empty((f.n, f.c, f.h, f.w), device=self.device, dtype=torch.float32)
            check(lib().mi355det_nhwc_to_nchw_f32(f.ptr, 1, f.ld, f.n, f.c, f.h, f.w, _vp(o), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "nhwc_to_nchw_f32")
            outs.append(o)
        return outs

    def feature_maps_nhwc(self, levels=4):
        """P2.. as the engine's own bf16 NHWC buffers (no copy): leaf tensors for `MultiScaleRoIAlign.forward_nhwc`."""
        return [f.buf.detach().requires_grad_(self.training) for f in self._last_plan.features[:levels]]

    def backward(self, grad_objectness, grad_deltas, feature_grads=None):
        """feature_grads: per level None | NCHW fp32 [n,C,h,w] | NHWC [n,h,w,C] (bf16 or fp32), e.g. the .grad of feature_maps_*()."""
        p = self._last_plan
        p.glogits.copy_(grad_objectness.reshape(p.glogits.shape))
        p.gbbox.copy_(grad_deltas.reshape(p.gbbox.shape))
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for l, rg in enumerate(p.roi_grads):
            g = None if feature_grads is None or l >= len(feature_grads) else feature_grads[l]
            if g is None:
                rg.buf.zero_()
            elif g.shape[-1] == rg.c and g.shape[1] == rg.h:          # channels-last
                rg.buf.copy_(g)
            else:
                g = g.float().contiguous()
                check(lib().mi355det_nchw_f32_to_nhwc(_vp(g), rg.n, rg.c, rg.h, rg.w, rg.ptr, 1, rg.ld, st), "nchw_f32_to_nhwc")
        p.load_head_grads()
        p.run_backward()


class RetinaPlan:
    """Buffers + prepared call lists for one (batch, H, W, mode)."""

    def __init__(self, eng, n, H, W, training, stream):
        self.eng, self.n, self.H, self.W, self.training = eng, n, H, W, training
        self.stream = C.c_void_p(stream)
        self.fwd, self.bwd, self.pack = [], [], []
        self.keep = []
        self.ops = []
        self.layers = {}
        dev, L, bf = eng.device, lib(), torch.bfloat16
        A, K = eng.na, eng.nc

        def new_act(n_, h_, w_, c_, needs_grad):
            a = Act(torch.zeros((n_, h_, w_, c_), device=dev, dtype=bf), n_, h_, w_, c_, c_)
            a.needs_grad = needs_grad
            a.parts = []
            return a

        # level geometry first: the heads write into level-concatenated outputs
        def down(v, times):
            for _ in range(times):
                v = (v - 1) // 2 + 1
            return v
        frcnn = eng.model == "fasterrcnn"
        sizes = [(down(H, t), down(W, t)) for t in ((2, 3, 4, 5, 6) if frcnn else (3, 4, 5, 6, 7))]
        self.level_sizes = sizes
        self.level_rows = [h * w * A for h, w in sizes]
        self.rows = sum(self.level_rows)
        self.logits = torch.zeros((n, self.rows, K), device=dev)
        self.bbox_reg = torch.zeros((n, self.rows, 4), device=dev)
        if training:
            self.glogits = torch.zeros_like(self.logits)
            self.gbbox = torch.zeros_like(self.bbox_reg)
            self.matched = torch.zeros((n, self.rows), dtype=torch.int64, device=dev)

        class IL:
            tensors = torch.empty((n, 3, H, W), device="meta")
            image_sizes = [(H, W)] * n
        self.anchors = eng.anchor_generator(IL, [torch.empty((1, 1, h, w), device=dev) for h, w in sizes])[0]
        self.anchors_per_level = list(self.anchors.split(self.level_rows))

        def conv(name, x, res=None, level=None):
            s = eng.by_name[name]
            shp_f = eng._shape(s, x.n, x.h, x.w, store=False, in_ld=x.ld)
            wf, wd = eng.packed[name]
            aff = eng.affine[s.bn] if s.bn else None
            scale = _vp(aff[0]) if aff is not None else None
            shift = _vp(aff[1]) if aff is not None else (_vp(eng.params[name + ".bias"]) if s.bias else None)
            needs = s.trainable or x.needs_grad or (res is not None and res.needs_grad)
            rec = dict(kind="conv", name=name, spec=s, x=x, res=res, level=level, scale=aff[0] if aff is not None else None)
            if s.head:
                k = K if s.head == "cls_logits" else 4
                out = self.logits if s.head == "cls_logits" else self.bbox_reg
                row0 = sum(self.level_rows[:level])
                shp_f.out_ld = A * k
                e = _lib.ConvEpilogue(None, shift, None, 0, 0, self.rows * k)
                y = C.c_void_p(out.data_ptr() + 4 * row0 * k)
                self.fwd.append((L.mi355det_conv_fwd_ex, (C.byref(shp_f), x.ptr, _vp(wf), C.byref(e), y, 1, ops.cout_pad_of(s.cout), self.stream)))
                a = None
            else:
                a = new_act(x.n, shp_f.ho, shp_f.wo, s.cout, needs)
                e = _lib.ConvEpilogue(scale, shift, res.ptr if res is not None else None, res.ld if res is not None else 0, int(s.relu), 0)
                self.fwd.append((L.mi355det_conv_fwd_ex, (C.byref(shp_f), x.ptr, _vp(wf), C.byref(e), a.ptr, 0, ops.cout_pad_of(s.cout), self.stream)))
            shp_b = eng._shape(s, x.n, x.h, x.w, store=True, in_ld=x.ld)          # backward view: padded cout, dense dz pitch
            self.keep += [shp_f, shp_b, e]
            rec.update(a=a, shp=shp_b, shp_f=shp_f)
            self.ops.append(rec)
            self.layers.setdefault(name, []).append(rec)
            return a

        # ---- stem: normalise + im2col (7x7/2) -> GEMM + FrozenBN + ReLU -> max-pool     (resnet.py:232-235)
        h2, w2 = down(H, 1), down(W, 1)
        s1 = eng.by_name["backbone.body.conv1"]
        self.img_call = len(self.fwd)
        if not s1.trainable and H % 32 == 0 and W % 32 == 0:
            # frozen stem (the reference default, backbone_utils.py:89-104): one direct 7x7/2 convolution kernel, no im2col matrix
            wf1, _ = eng.packed["backbone.body.conv1"]
            aff1 = eng.affine[s1.bn]
            c1 = new_act(n, h2, w2, 64, False)
            self.fwd.append((L.mi355det_resnet_stem_fwd, [None, _vp(eng.mean) if eng.normalize else None, _vp(eng.inv_std) if eng.normalize else None,
                                                          _vp(wf1), _vp(aff1[0]), _vp(aff1[1]), int(s1.relu), c1.ptr, c1.ld, n, H, W, self.stream]))
            self.ops.append(dict(kind="rstem", a=c1))
        else:
            self.col = new_act(n, h2, w2, STEM_K, False)
            self.fwd.append((L.mi355det_im2col_nchw, [None, _vp(eng.mean) if eng.normalize else None, _vp(eng.inv_std) if eng.normalize else None,
                                                      self.col.ptr, n, 3, H, W, 7, 2, 3, STEM_K, self.stream]))
            c1 = conv("backbone.body.conv1", self.col)
        x = new_act(n, down(h2, 1), down(w2, 1), 64, c1.needs_grad)       # only a trained stem needs the gradient through the max-pool
        self.fwd.append((L.mi355det_maxpool3x3s2, (c1.ptr, c1.ld, n, c1.h, c1.w, 64, x.ptr, x.ld, self.stream)))
        self.ops.append(dict(kind="maxpool", x=c1, a=x))
        feats = []
        for li, nb in enumerate(BODY_LAYERS[eng.body_name], 1):
            for b in range(nb):
                q = f"backbone.body.layer{li}.{b}"
                idn = conv(q + ".downsample.0", x) if b == 0 else x
                y = conv(q + ".conv1", x)
                y = conv(q + ".conv2", y)
                x = conv(q + ".conv3", y, res=idn)
            feats.append(x)
        self.body = feats
        # ---- FPN (top-down) + extra levels
        Fp = "backbone.fpn."
        c_feats = feats if frcnn else feats[1:]          # Faster R-CNN returns C2..C5, RetinaNet C3..C5
        top = len(c_feats) - 1
        last = conv(f"{Fp}inner_blocks.{top}", c_feats[top])
        outs = [conv(f"{Fp}layer_blocks.{top}", last)]
        for i in range(top - 1, -1, -1):
            lat = conv(f"{Fp}inner_blocks.{i}", c_feats[i])
            merged = new_act(n, lat.h, lat.w, 256, True)
            self.fwd.append((L.mi355det_upsample_nearest_add, (last.ptr, last.ld, n, last.h, last.w, 256, lat.ptr, lat.ld, lat.h, lat.w,
                                                               merged.ptr, merged.ld, self.stream)))
            self.ops.append(dict(kind="up_add", top=last, lat=lat, a=merged))
            last = merged
            outs.insert(0, conv(f"{Fp}layer_blocks.{i}", last))
        if frcnn:
            # LastLevelMaxPool: F.max_pool2d(P5, 1, 2, 0) = every other pixel (torchvision feature_pyramid_network); the RPN reads it,
            # the RoI heads do not.  A strided torch copy inside the call list (tiny map).
            p5 = outs[-1]
            pool = new_act(n, down(p5.h, 1), down(p5.w, 1), 256, True)
            self.fwd.append((comm_hook, (lambda: pool.buf.copy_(p5.buf[:, ::2, ::2]),)))
            self.ops.append(dict(kind="pool", x=p5, a=pool))
            self.features = outs + [pool]
        else:
            p6 = conv(Fp + "extra_blocks.p6", outs[-1])
            r6 = new_act(n, p6.h, p6.w, 256, True)
            # F.relu(p6) (LastLevelP6P7.forward) on a tiny map: the BN+activation kernel with unit scale, zero shift, slope 0
            self.unit_ss = torch.zeros(4 * 256, device=dev)
            self.unit_ss[:256] = 1.0
            self.fwd.append((L.mi355det_bn_act_fwd, (p6.ptr, p6.ld, _vp(self.unit_ss), 256, p6.pixels, 0.0, None, 0, r6.ptr, r6.ld, self.stream)))
            self.ops.append(dict(kind="relu", x=p6, a=r6))
            p7 = conv(Fp + "extra_blocks.p7", r6)
            self.features = outs + [p6, p7]
        assert [(f.h, f.w) for f in self.features] == sizes, ([(f.h, f.w) for f in self.features], sizes)
        if frcnn:
            # gradients arriving from the RoI heads (RoIAlign backward on P2..P5) enter the backward pass as extra contributions
            self.roi_grads = []
            for f in self.features[:4]:
                gbuf = new_act(n, f.h, f.w, 256, False)
                self.roi_grads.append(gbuf)
            # ---- RPN head shared over the five levels (rpn.py:17-58)
            for lvl, f in enumerate(self.features):
                t = conv("rpn.head.conv", f)
                conv("rpn.head.cls_logits", t, level=lvl)
                conv("rpn.head.bbox_pred", t, level=lvl)
        else:
            # ---- heads, weights shared over the five levels (retinanet.py:150-170,228-246)
            for hname, last_name in (("classification_head", "cls_logits"), ("regression_head", "bbox_reg")):
                for lvl, f in enumerate(self.features):
                    t = f
                    for i in (0, 2, 4, 6):
                        t = conv(f"head.{hname}.conv.{i}", t)
                    conv(f"head.{hname}.{last_name}", t, level=lvl)

        # ---- weight packing of the trainable convolutions (every step: the optimizer changes the fp32 masters)
        tr = [s for s in eng.specs if s.trainable]
        if tr:
            self.pack_table, ne, nb = eng._pack_table(tr, need_dgrad=training)
            self.pack.append((L.mi355det_pack_weights_batched, (_vp(self.pack_table), ne, nb, self.stream)))
        if training:
            self._build_backward()
            self._autotune()
        else:
            self._autotune_eval()

    # ------------------------------------------------------------------
    def _build_backward(self):
        eng, L, dev, bf = self.eng, lib(), self.eng.device, torch.bfloat16
        A, K, n = eng.na, eng.nc, self.n
        # bf16 head-gradient buffers per level (padded channel pitch), filled from the fp32 loss gradients by cast_rows
        self.head_grads = {}
        self.cast, self.cast_box = [], []
        for lvl, (h, w) in enumerate(self.level_sizes):
            row0 = sum(self.level_rows[:lvl])
            for key, src, k in (("cls_logits", self.glogits, K), ("bbox_reg", self.gbbox, 4)):
                ld = ops.pad_to(A * k, 64)          # = Conv.cout_store of the head convolutions
                gbuf = torch.zeros((n, h, w, ld), device=dev, dtype=bf)
                self.head_grads[(key, lvl)] = gbuf
                call = (L.mi355det_cast_rows_bf16, (C.c_void_p(src.data_ptr() + 4 * row0 * k), self.rows * k, A * k, n, h * w, A * k, 1.0,
                                                    _vp(gbuf), ld, self.stream))
                self.cast.append(call)
                if key == "bbox_reg":
                    self.cast_box.append(call)
        dz_elems = max(r["shp"].n * r["shp"].ho * r["shp"].wo * r["shp"].cout for r in self.ops if r["kind"] == "conv")
        self.dz2 = [torch.zeros(dz_elems, device=dev, dtype=bf) for _ in range(2)]
        self.side = torch.cuda.Stream(device=dev)
        side_ptr = C.c_void_p(self.side.cuda_stream)
        main = torch.cuda.current_stream(dev)
        wg_done = [None, None]
        flip = [0]
        ws_need = max(L.mi355det_conv_wgrad_workspace(C.byref(r["shp"])) for r in self.ops if r["kind"] == "conv")
        self.wgrad_ws = torch.empty(max(ws_need, 16), device=dev, dtype=torch.uint8)
        ws_ptr, ws_bytes = _vp(self.wgrad_ws), self.wgrad_ws.numel()

        def py(fn, *a):
            self.bwd.append((comm_hook, (fn,) + a))

        # Every gradient buffer is OWNED by the plan (self.grad_bufs): the call list bakes raw device pointers, and a buffer that was only
        # reachable through an activation's `parts` queue was freed as soon as the queue handed it to a call (residual of a data gradient,
        # operand of an add) - the caching allocator then gave the block to whoever asked next (round 4 found the Faster R-CNN box head's
        # weight packs, created in the first training call, overwritten by every later backward: tests/test_gpu_fullsize_tv.py).
        self.grad_bufs = []

        def dense(a):
            g = Act(torch.zeros((a.n, a.h, a.w, a.c), device=dev, dtype=bf), a.n, a.h, a.w, a.c, a.c)
            self.grad_bufs.append(g.buf)
            return g

        # ---- gradient bookkeeping: an activation's gradient is the sum of its consumers' contributions; tensors that already
        #      exist are queued as `parts` so that the first data-gradient GEMM can take one as its epilogue residual
        def add_tensor(x, t):
            if not x.needs_grad:
                return
            if x.grad_written:
                self.bwd.append((L.mi355det_add_bf16, (x.grad.ptr, x.grad.ld, t.ptr, t.ld, x.c, x.pixels, x.grad.ptr, x.grad.ld, self.stream)))
            else:
                x.parts.append(t)

        # split-K data gradients (few pixels, deep reduction: the LVIS cls_logits on the small levels) share one fp32 workspace; they run
        # one after the other on the main stream
        dws_need = max(L.mi355det_conv_dgrad_workspace(C.byref(r["shp"])) for r in self.ops if r["kind"] == "conv")
        self.dgrad_ws = torch.empty(max(dws_need, 16), device=dev, dtype=torch.uint8) if dws_need else None

        def dgrad_call(shp, dy_ptr, wd, g, rptr, rld):
            if self.dgrad_ws is not None and L.mi355det_conv_dgrad_workspace(C.byref(shp)):
                return (L.mi355det_conv_dgrad_ws, (C.byref(shp), dy_ptr, _vp(wd), g.ptr, rptr, rld, _vp(self.dgrad_ws), self.dgrad_ws.numel(), self.stream))
            return (L.mi355det_conv_dgrad, (C.byref(shp), dy_ptr, _vp(wd), g.ptr, rptr, rld, self.stream))

        # ---- FrozenBN / ReLU backward folded into the data gradient that produces its input (mi355det_conv_dgrad_mask): possible when the
        #      activation has exactly ONE consumer, a stride-1 convolution, and its producer is a plain conv (+affine) (+ReLU) without a
        #      residual - conv1 -> conv2 -> conv3 inside a bottleneck, the head towers, FPN laterals (-0.2 ... -0.5 ms per step, profiles/r03_ab_results.md).
        uses = {}
        for r in self.ops:
            for key in ("x", "res", "lat", "top"):
                t = r.get(key)
                if t is not None:
                    uses[id(t)] = uses.get(id(t), 0) + 1
        for f in self.features:                        # the heads (and, Faster R-CNN, RoIAlign) read the pyramid levels as well
            uses[id(f)] = uses.get(id(f), 0) + (1 if getattr(self, "roi_grads", None) else 0)
        producer = {id(r["a"]): r for r in self.ops if r["kind"] == "conv" and r.get("a") is not None}
        fuse_ok = True

        def mask_fusable(x, shp):
            pr = producer.get(id(x))
            if not fuse_ok or pr is None or uses.get(id(x), 0) != 1 or shp.stride != 1 or x.parts or x.grad_written:
                return None
            ps = pr["spec"]
            if ps.head or pr["res"] is not None or not (ps.relu or pr["scale"] is not None):
                return None
            if not (ps.trainable or pr["x"].needs_grad):
                return None
            if self.dgrad_ws is not None and L.mi355det_conv_dgrad_workspace(C.byref(shp)):
                return None                            # the split-K form keeps its own epilogue
            return pr

        def add_dgrad(x, shp, dy_ptr, wd):
            if not x.needs_grad:
                return
            if x.grad is None:
                x.grad = dense(x)
            g = x.grad
            pr = mask_fusable(x, shp)
            if pr is not None:
                self.bwd.append((L.mi355det_conv_dgrad_mask, (C.byref(shp), dy_ptr, _vp(wd), g.ptr, x.ptr, x.ld, _vp(pr["scale"]), int(pr["spec"].relu),
                                                              self.stream)))
                x.grad_written = True
                pr["dz_fused"] = True                  # x.grad now holds dz of the producer, not the activation gradient
                return
            if x.grad_written:
                self.bwd.append(dgrad_call(shp, dy_ptr, wd, g, g.ptr, g.ld))
            else:
                r = x.parts.pop(0) if x.parts else None
                self.bwd.append(dgrad_call(shp, dy_ptr, wd, g, r.ptr if r else None, r.ld if r else 0))
                x.grad_written = True
                while x.parts:
                    t = x.parts.pop(0)
                    self.bwd.append((L.mi355det_add_bf16, (g.ptr, g.ld, t.ptr, t.ld, x.c, x.pixels, g.ptr, g.ld, self.stream)))

        def finalize(a):
            """Gradient of `a` once every consumer has contributed (ops are walked in reverse)."""
            if a.grad_written:
                return a.grad
            if not a.parts:
                return None
            if len(a.parts) == 1:
                a.grad = a.parts.pop(0)          # alias, no copy
            else:
                a.grad = dense(a)
                p0, p1 = a.parts.pop(0), a.parts.pop(0)
                self.bwd.append((L.mi355det_add_bf16, (p0.ptr, p0.ld, p1.ptr, p1.ld, a.c, a.pixels, a.grad.ptr, a.grad.ld, self.stream)))
                while a.parts:
                    t = a.parts.pop(0)
                    self.bwd.append((L.mi355det_add_bf16, (a.grad.ptr, a.grad.ld, t.ptr, t.ld, a.c, a.pixels, a.grad.ptr, a.grad.ld, self.stream)))
            a.grad_written = True
            return a.grad

        self.bwd_marks = []
        first_off = {name: o for name, o, _n, _s in eng.param_order}
        for f, rg in zip(self.features, getattr(self, "roi_grads", [])):
            f.parts.append(rg)                 # Faster R-CNN: gradient of P2..P5 coming back through RoIAlign
        for rec in reversed(self.ops):
            kind = rec["kind"]
            if kind == "rstem":                # frozen stem: nothing flows back through it
                continue
            if kind == "pool":
                g = finalize(rec["a"])
                if g is None:
                    continue
                x = rec["x"]
                d = dense(x)                   # zeros except the sampled pixels
                self.bwd.append((comm_hook, ((lambda d=d, g=g: d.buf[:, ::2, ::2].copy_(g.buf)),)))
                add_tensor(x, d)
                continue
            if kind == "maxpool":
                g = finalize(rec["a"])
                if g is None:
                    continue
                x = rec["x"]
                d = dense(x)
                self.bwd.append((L.mi355det_maxpool3x3s2_bwd, (x.ptr, x.ld, g.ptr, g.ld, x.n, x.h, x.w, x.c, d.ptr, d.ld, self.stream)))
                add_tensor(x, d)
                continue
            if kind == "up_add":
                g = finalize(rec["a"])
                if g is None:
                    continue
                add_tensor(rec["lat"], g)
                top = rec["top"]
                if top.needs_grad:
                    if top.grad is None:
                        top.grad = dense(top)
                    acc = top.grad if top.grad_written else (top.parts.pop(0) if top.parts else None)
                    self.bwd.append((L.mi355det_upsample_nearest_bwd, (g.ptr, g.ld, top.n, top.h, top.w, top.c, g.h, g.w, acc.ptr if acc else None,
                                                                       acc.ld if acc else 0, top.grad.ptr, top.grad.ld, self.stream)))
                    top.grad_written = True
                    while top.parts:
                        t = top.parts.pop(0)
                        self.bwd.append((L.mi355det_add_bf16, (top.grad.ptr, top.grad.ld, t.ptr, t.ld, top.c, top.pixels, top.grad.ptr, top.grad.ld,
                                                               self.stream)))
                continue
            if kind == "relu":
                g = finalize(rec["a"])
                if g is None:
                    continue
                a, x = rec["a"], rec["x"]
                d = dense(a)
                self.bwd.append((L.mi355det_relu_affine_bwd, (g.ptr, g.ld, None, 0, a.ptr, a.ld, None, a.c, a.pixels, 1, d.ptr, d.ld, None, 0,
                                                              self.stream)))
                add_tensor(x, d)
                continue
            s, x, a, res, shp, name = rec["spec"], rec["x"], rec["a"], rec["res"], rec["shp"], rec["name"]
            if s.head:
                gbuf = self.head_grads[(s.head, rec["level"])]
                dy_ptr, fresh_dz = _vp(gbuf), None
            else:
                g = finalize(a)
                if g is None:
                    continue
                need_dz = s.trainable or x.needs_grad
                need_gm = res is not None and res.needs_grad
                if not (need_dz or need_gm):
                    continue
                scale = rec["scale"]
                if rec.get("dz_fused"):
                    dy_ptr, fresh_dz = g.ptr, None         # the consumer's data gradient already applied scale and mask: g IS dz
                elif s.relu or scale is not None:
                    gm = dense(a) if need_gm else None
                    fresh_dz = None
                    if need_dz:
                        fresh_dz = flip[0]
                        flip[0] ^= 1
                        if wg_done[fresh_dz] is not None:
                            py(main.wait_event, wg_done[fresh_dz])
                    dzp = _vp(self.dz2[fresh_dz]) if need_dz else None
                    self.bwd.append((L.mi355det_relu_affine_bwd, (g.ptr, g.ld, None, 0, a.ptr, a.ld, _vp(scale), a.c, a.pixels, int(s.relu), dzp, a.c,
                                                                  gm.ptr if gm else None, gm.ld if gm else 0, self.stream)))
                    if need_gm:
                        add_tensor(res, gm)
                    dy_ptr = dzp
                else:
                    dy_ptr, fresh_dz = g.ptr, None        # plain bias conv: dz is the incoming gradient itself
                    if need_gm:
                        add_tensor(res, g)
                if not need_dz:
                    continue
            rec["dy_ptr"] = dy_ptr
            if s.trainable:
                ev_dz, ev_wg = torch.cuda.Event(), torch.cuda.Event()
                py(ev_dz.record, main)
                py(self.side.wait_event, ev_dz)
                self.bwd.append((L.mi355det_conv_wgrad, (C.byref(shp), x.ptr, dy_ptr, _vp(eng.grads[name + ".weight"]),
                                                         _vp(eng.grads[name + ".bias"]) if s.bias else None, ws_ptr, ws_bytes, side_ptr)))
                py(ev_wg.record, self.side)
                if fresh_dz is not None:
                    wg_done[fresh_dz] = ev_wg
            if x.needs_grad:
                _, wd = eng.packed[name]
                add_dgrad(x, shp, dy_ptr, wd)
            if s.trainable:
                self.bwd_marks.append((len(self.bwd), first_off[name + ".weight"]))
        ev_end = torch.cuda.Event()
        py(ev_end.record, self.side)
        py(main.wait_event, ev_end)
        self.side_stream = self.side
        # DDP bucket marks (position in self.bwd after which flat_g[offset:] is final).  The head weights are shared by the five
        # levels, so a parameter is final only after the LAST launch that touches it.
        last = {}
        for pos, off in self.bwd_marks:
            last[off] = max(last.get(off, 0), pos)
        marks, mx = [], 0
        for off in sorted(last, reverse=True):
            mx = max(mx, last[off])
            marks.append((mx, off))
        self.bwd_marks = marks

    def _autotune_eval(self):
        """Inference plans time the tile candidates of their forward launches too (the igemm tuning key includes the epilogue and the
        pixel count: an eval plan shares nothing with a training plan of another batch size).  Before: every convolution of a RetinaNet
        inference ran the default 128x128 tile - cls_logits of the 1204-class head 6.6 ms against 4.9 ms tuned."""
        eng, L = self.eng, lib()
        img = torch.rand((self.n, 3, self.H, self.W), device=eng.device)
        self.fwd[self.img_call][1][0] = C.c_void_p(img.data_ptr())
        L.mi355det_conv_autotune_mode(1)
        try:
            self._run(self.pack)
            self._run(self.fwd)
        finally:
            L.mi355det_conv_autotune_mode(0)
        torch.cuda.synchronize()

    def _autotune(self):
        eng, L = self.eng, lib()
        img = torch.rand((self.n, 3, self.H, self.W), device=eng.device)
        self.fwd[self.img_call][1][0] = C.c_void_p(img.data_ptr())
        self.glogits.normal_(0, 1e-3)
        self.gbbox.normal_(0, 1e-3)
        L.mi355det_conv_autotune_mode(1)
        try:
            self._run(self.pack)
            self._run(self.fwd)
            self._run(self.cast)
            self.side.wait_stream(torch.cuda.current_stream())
            self._run(self.bwd)
        finally:
            L.mi355det_conv_autotune_mode(0)
        torch.cuda.synchronize()
        ws_ptr, ws_bytes = _vp(self.wgrad_ws), self.wgrad_ws.numel()
        for rec in self.ops:
            if rec["kind"] == "conv" and rec["spec"].trainable and "dy_ptr" in rec:
                st = L.mi355det_conv_wgrad_autotune(C.byref(rec["shp"]), rec["x"].ptr, rec["dy_ptr"], _vp(eng.grads[rec["name"] + ".weight"]),
                                                    ws_ptr, ws_bytes, self.stream)
                if st < 0:
                    check(st, "conv_wgrad_autotune")
        torch.cuda.synchronize()
        eng.flat_g.zero_()
        self.glogits.zero_()
        self.gbbox.zero_()

    # ------------------------------------------------------------------
    def _run(self, calls):
        for fn, args in calls:
            if fn is comm_hook:
                args[0](*args[1:])
                continue
            st = fn(*args)
            if st != 0:
                check(st, fn.__name__)

    def run_forward(self, images):
        self._img = images
        self.fwd[self.img_call][1][0] = C.c_void_p(images.data_ptr())
        self._run(self.pack)
        self._run(self.fwd)

    def head_gradient(self, key="cls_logits"):
        """The bf16 head gradient the backward consumes, re-assembled as fp32 [n, rows, k] (tests / diagnostics)."""
        A = self.eng.na
        k = self.eng.nc if key == "cls_logits" else 4
        parts = [self.head_grads[(key, lvl)][..., :A * k].float().reshape(self.n, -1, k) for lvl in range(len(self.level_sizes))]
        return torch.cat(parts, dim=1)

    def load_head_grads(self, cls=True):
        self._run(self.cast if cls else self.cast_box)

    def run_backward(self):
        self.eng.flat_g.zero_()
        self.side.wait_stream(torch.cuda.current_stream())
        self._run(self.bwd)
