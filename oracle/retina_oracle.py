"""Plain PyTorch fp32 restatement of the reference RetinaNet-ResNet50-FPN network — TEST INFRASTRUCTURE ONLY.

  * ResNet-50 body (torchvision_models/utilities/resnet.py:87-143 Bottleneck, :173-176,:230-240 stem + layers) with
    FrozenBatchNorm2d (tvision/backbone_utils.py:67: norm_layer=misc_nn_ops.FrozenBatchNorm2d) — pinned by
    tests/golden/g12_retinanet.npz (outputs of the reference's own ResNet class on deterministic weights);
  * RetinaNetHead (tvision/retinanet.py:66-105,150-170,173-246) — pinned by the same fixture (reference modules);
  * FeaturePyramidNetwork + LastLevelP6P7 (torchvision.ops.feature_pyramid_network, NOT vendored in the reference and
    torchvision is not installed): restated from the published semantics, **parity unpinned**.

Functional graph over a flat {name: tensor} state dict whose keys and order equal
`retinanet_resnet50_fpn(...).state_dict()` minus the transform.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import detrand

LAYERS = [3, 4, 6, 3]
PLANES = [64, 128, 256, 512]
FROZEN_BN_EPS = 1e-5          # torchvision FrozenBatchNorm2d default (overwrite_eps(model, 0.0) only with pretrained weights)
IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
ANCHOR_SIZES = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in [32, 64, 128, 256, 512])   # retinanet.py:647
ASPECT_RATIOS = ((0.5, 1.0, 2.0),) * 5


BODY_LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}      # utilities/resnet.py:292-313


def body_keys(prefix="backbone.body.", layers=None):
    layers = layers or LAYERS
    out = []

    def conv(name, cout, cin, k):
        out.append((prefix + name + ".weight", (cout, cin, k, k)))

    def bn(name, c):
        for s in (".weight", ".bias", ".running_mean", ".running_var"):
            out.append((prefix + name + s, (c,)))
    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    inpl = 64
    for li, (planes, nb) in enumerate(zip(PLANES, layers), 1):
        for b in range(nb):
            q = f"layer{li}.{b}"
            conv(q + ".conv1", planes, inpl, 1)
            bn(q + ".bn1", planes)
            conv(q + ".conv2", planes, planes, 3)
            bn(q + ".bn2", planes)
            conv(q + ".conv3", planes * 4, planes, 1)
            bn(q + ".bn3", planes * 4)
            if b == 0:
                conv(q + ".downsample.0", planes * 4, inpl, 1)
                bn(q + ".downsample.1", planes * 4)
            inpl = planes * 4
    return out


def fpn_keys(prefix="backbone.fpn."):
    out = []
    for name, chans in (("inner_blocks", [(512, 1), (1024, 1), (2048, 1)]), ("layer_blocks", [(256, 3)] * 3)):
        for i, (cin, k) in enumerate(chans):
            out.append((f"{prefix}{name}.{i}.weight", (256, cin, k, k)))
            out.append((f"{prefix}{name}.{i}.bias", (256,)))
    for p in ("p6", "p7"):
        out.append((f"{prefix}extra_blocks.{p}.weight", (256, 256, 3, 3)))
        out.append((f"{prefix}extra_blocks.{p}.bias", (256,)))
    return out


def head_keys(num_classes=91, num_anchors=9, prefix="head."):
    out = []
    for hname, last, cout in (("classification_head", "cls_logits", num_anchors * num_classes), ("regression_head", "bbox_reg", num_anchors * 4)):
        for i in (0, 2, 4, 6):
            out.append((f"{prefix}{hname}.conv.{i}.weight", (256, 256, 3, 3)))
            out.append((f"{prefix}{hname}.conv.{i}.bias", (256,)))
        out.append((f"{prefix}{hname}.{last}.weight", (cout, 256, 3, 3)))
        out.append((f"{prefix}{hname}.{last}.bias", (cout,)))
    return out


def state_keys(num_classes=91, num_anchors=9, body="resnet50"):
    """`body` = "resnet101" composes BASELINE config 5 (resnet_fpn_backbone('resnet101', ...) + RetinaNet(backbone, 1204))."""
    return body_keys(layers=BODY_LAYERS[body]) + fpn_keys() + head_keys(num_classes, num_anchors)


def det_fill(key, shape, seed):
    """Deterministic value for one state-dict entry (shared with tools/make_golden.py, which applies it to the reference's
    own modules).  Scales keep activations O(1) through 16 bottlenecks: He-uniform convolutions, damped bn3."""
    if key.endswith("running_mean"):
        return detrand.uniform(seed, shape, -0.2, 0.2)
    if key.endswith("running_var"):
        return detrand.uniform(seed, shape, 0.5, 1.5)
    is_bn = ".bn" in key or ".downsample.1." in key
    if is_bn and key.endswith("weight"):
        return detrand.uniform(seed, shape, 0.15, 0.35) if ".bn3." in key else detrand.uniform(seed, shape, 0.5, 1.5)
    if is_bn and key.endswith("bias"):
        return detrand.uniform(seed, shape, -0.2, 0.2)
    if key.endswith("bias"):
        if key.endswith("cls_logits.bias"):
            return np.full(shape, -math.log(99.0), np.float32)          # retinanet.py:97 prior probability 0.01
        return detrand.uniform(seed, shape, -0.1, 0.1)
    fan = shape[1] * shape[2] * shape[3]
    s = math.sqrt(3.0) * math.sqrt(2.0 / fan)
    if ".cls_logits." in key or ".bbox_reg." in key:
        s *= 0.25
    return detrand.uniform(seed, shape, -s, s)


def det_state(seed, num_classes=91, num_anchors=9, keys=None, body="resnet50"):
    sd = {}
    for i, (k, shp) in enumerate(keys or state_keys(num_classes, num_anchors, body)):
        sd[k] = torch.from_numpy(np.ascontiguousarray(det_fill(k, shp, seed + i)))
    return sd


def frozen_bn(x, sd, name, eps=FROZEN_BN_EPS):
    """torchvision.ops.misc.FrozenBatchNorm2d.forward: scale = w * rsqrt(rv + eps); y = x*scale + (b - rm*scale)."""
    w, b, rm, rv = (sd[name + s].view(1, -1, 1, 1) for s in (".weight", ".bias", ".running_mean", ".running_var"))
    scale = w * (rv + eps).rsqrt()
    return x * scale + (b - rm * scale)


def bottleneck(x, sd, q, stride):
    """utilities/resnet.py:122-143 (stride on the 3x3: ResNet v1.5)."""
    out = F.relu(frozen_bn(F.conv2d(x, sd[q + ".conv1.weight"]), sd, q + ".bn1"))
    out = F.relu(frozen_bn(F.conv2d(out, sd[q + ".conv2.weight"], stride=stride, padding=1), sd, q + ".bn2"))
    out = frozen_bn(F.conv2d(out, sd[q + ".conv3.weight"]), sd, q + ".bn3")
    idn = x
    if q + ".downsample.0.weight" in sd:
        idn = frozen_bn(F.conv2d(x, sd[q + ".downsample.0.weight"], stride=stride), sd, q + ".downsample.1")
    return F.relu(out + idn)


def body_forward(sd, x, prefix="backbone.body."):
    """-> [C2, C3, C4, C5] (layer1..layer4 outputs), utilities/resnet.py:230-240."""
    x = F.relu(frozen_bn(F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3), sd, prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li in range(1, 5):
        b = 0
        while f"{prefix}layer{li}.{b}.conv1.weight" in sd:      # block counts from the state dict: 3/4/6/3 (R50), 3/4/23/3 (R101)
            x = bottleneck(x, sd, f"{prefix}layer{li}.{b}", 2 if (b == 0 and li > 1) else 1)
            b += 1
        feats.append(x)
    return feats


def fpn_forward(sd, feats, prefix="backbone.fpn."):
    """FeaturePyramidNetwork(in=[512,1024,2048], out=256, extra=LastLevelP6P7(256,256)) on [C3,C4,C5] -> [P3..P7]."""
    def inner(i, t):
        return F.conv2d(t, sd[f"{prefix}inner_blocks.{i}.weight"], sd[f"{prefix}inner_blocks.{i}.bias"])

    def layer(i, t):
        return F.conv2d(t, sd[f"{prefix}layer_blocks.{i}.weight"], sd[f"{prefix}layer_blocks.{i}.bias"], padding=1)
    last = inner(2, feats[2])
    outs = [layer(2, last)]
    for i in (1, 0):
        lat = inner(i, feats[i])
        last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
        outs.insert(0, layer(i, last))
    p6 = F.conv2d(outs[-1], sd[prefix + "extra_blocks.p6.weight"], sd[prefix + "extra_blocks.p6.bias"], stride=2, padding=1)   # in==out: uses P5
    p7 = F.conv2d(F.relu(p6), sd[prefix + "extra_blocks.p7.weight"], sd[prefix + "extra_blocks.p7.bias"], stride=2, padding=1)
    return outs + [p6, p7]


def head_forward(sd, features, num_classes=91, prefix="head."):
    """RetinaNetHead.forward (retinanet.py:150-170,228-246): -> cls_logits [N, sum HWA, K], bbox_regression [N, sum HWA, 4]."""
    res = []
    for hname, last, k in (("classification_head", "cls_logits", num_classes), ("regression_head", "bbox_reg", 4)):
        outs = []
        for f in features:
            t = f
            for i in (0, 2, 4, 6):
                t = F.relu(F.conv2d(t, sd[f"{prefix}{hname}.conv.{i}.weight"], sd[f"{prefix}{hname}.conv.{i}.bias"], padding=1))
            t = F.conv2d(t, sd[f"{prefix}{hname}.{last}.weight"], sd[f"{prefix}{hname}.{last}.bias"], padding=1)
            n, _, h, w = t.shape
            outs.append(t.view(n, -1, k, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, k))
        res.append(torch.cat(outs, 1))
    return res[0], res[1]


def normalize(images, mean=IMAGE_MEAN, std=IMAGE_STD):
    """GeneralizedRCNNTransform.normalize (tvision/transform.py:120-124)."""
    m = torch.tensor(mean, dtype=images.dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=images.dtype).view(1, 3, 1, 1)
    return (images - m) / s


def forward(sd, images, num_classes=91, do_normalize=True):
    x = normalize(images) if do_normalize else images
    body = body_forward(sd, x)
    feats = fpn_forward(sd, body[1:])
    cls_logits, bbox_reg = head_forward(sd, feats, num_classes)
    return {"body": body, "features": feats, "cls_logits": cls_logits, "bbox_regression": bbox_reg}


def sample(t, count=64):
    """Fixed strided subsample of a tensor (fixtures stay small)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // count)
    return f[::step][:count].numpy().copy()


# ---- Faster R-CNN backbone side: FPN over C2..C5 + LastLevelMaxPool, RPNHead (tvision/rpn.py:17-58; torchvision FPN: unpinned) ----
def frcnn_state_keys():
    out = body_keys()
    P = "backbone.fpn."
    for name, chans in (("inner_blocks", [(256, 1), (512, 1), (1024, 1), (2048, 1)]), ("layer_blocks", [(256, 3)] * 4)):
        for i, (cin, k) in enumerate(chans):
            out.append((f"{P}{name}.{i}.weight", (256, cin, k, k)))
            out.append((f"{P}{name}.{i}.bias", (256,)))
    out += [("rpn.head.conv.weight", (256, 256, 3, 3)), ("rpn.head.conv.bias", (256,)), ("rpn.head.cls_logits.weight", (3, 256, 1, 1)),
            ("rpn.head.cls_logits.bias", (3,)), ("rpn.head.bbox_pred.weight", (12, 256, 1, 1)), ("rpn.head.bbox_pred.bias", (12,))]
    return out


def frcnn_forward(sd, images, do_normalize=True):
    """-> {'features': [P2, P3, P4, P5, pool], 'objectness': [N, sum HWA, 1], 'deltas': [N, sum HWA, 4]} in the layout of
    rpn.py:concat_box_prediction_layers."""
    x = normalize(images) if do_normalize else images
    body = body_forward(sd, x)
    P = "backbone.fpn."

    def inner(i, t):
        return F.conv2d(t, sd[f"{P}inner_blocks.{i}.weight"], sd[f"{P}inner_blocks.{i}.bias"])

    def layer(i, t):
        return F.conv2d(t, sd[f"{P}layer_blocks.{i}.weight"], sd[f"{P}layer_blocks.{i}.bias"], padding=1)
    last = inner(3, body[3])
    outs = [layer(3, last)]
    for i in (2, 1, 0):
        lat = inner(i, body[i])
        last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
        outs.insert(0, layer(i, last))
    feats = outs + [F.max_pool2d(outs[-1], 1, 2, 0)]
    obj, dl = [], []
    for f in feats:
        t = F.relu(F.conv2d(f, sd["rpn.head.conv.weight"], sd["rpn.head.conv.bias"], padding=1))
        o = F.conv2d(t, sd["rpn.head.cls_logits.weight"], sd["rpn.head.cls_logits.bias"])
        d = F.conv2d(t, sd["rpn.head.bbox_pred.weight"], sd["rpn.head.bbox_pred.bias"])
        n, _, h, w = o.shape
        obj.append(o.view(n, -1, 1, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, 1))
        dl.append(d.view(n, -1, 4, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, 4))
    return {"body": body, "features": feats, "objectness": torch.cat(obj, 1), "deltas": torch.cat(dl, 1)}
