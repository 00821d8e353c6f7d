"""Plain PyTorch fp32 restatement of the reference network — TEST INFRASTRUCTURE ONLY.

DarkNet backbone (yolo/nets/backbone/darknet.py:10-107) + YoloHead (yolo/nets/yolohead.py:14-88)
as a functional graph over a flat {name: tensor} state dict whose keys and order equal the
reference module's state_dict().  Pinned by tests/golden/g8_network.npz (outputs / gradients of
the reference modules themselves on deterministic weights).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import detrand

BLOCKS = {"darknet_21": [1, 1, 2, 2, 1], "darknet_53": [1, 2, 8, 8, 4]}


def _bn_keys(prefix):
    return [prefix + s for s in (".weight", ".bias", ".running_mean", ".running_var", ".num_batches_tracked")]


def state_keys(backbone, na=3, nc=80):
    """(key, shape) list in the reference's state_dict order."""
    out = []

    def conv(name, cout, cin, k, bias=False):
        out.append((name + ".weight", (cout, cin, k, k)))
        if bias:
            out.append((name + ".bias", (cout,)))

    def bn(name, c):
        for kname in _bn_keys(name):
            out.append((kname, () if kname.endswith("tracked") else (c,)))
    conv("backbone.conv1", 32, 3, 3)
    bn("backbone.bn1", 32)
    inpl = 32
    for li, (planes, nb) in enumerate(zip([(32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)], BLOCKS[backbone]), 1):
        p = f"backbone.layer{li}"
        conv(p + ".ds_conv", planes[1], inpl, 3)
        bn(p + ".ds_bn", planes[1])
        inpl = planes[1]
        for b in range(nb):
            q = f"{p}.residual_{b}"
            conv(q + ".conv1", planes[0], inpl, 1)
            bn(q + ".bn1", planes[0])
            conv(q + ".conv2", planes[1], planes[0], 3)
            bn(q + ".bn2", planes[1])
    fo = na * (5 + nc)

    def cbl(name, cin, cout, k):
        conv(name + ".conv", cout, cin, k)
        bn(name + ".bn", cout)

    def emb(name, fl, cin):
        chans = [(cin, fl[0], 1), (fl[0], fl[1], 3), (fl[1], fl[0], 1), (fl[0], fl[1], 3), (fl[1], fl[0], 1), (fl[0], fl[1], 3)]
        for i, (ci, co, k) in enumerate(chans):
            cbl(f"{name}.{i}", ci, co, k)
        conv(name + ".conv_out", fo, fl[1], 1, bias=True)
    emb("embedding0", (512, 1024), 1024)
    cbl("embedding1_cbl", 512, 256, 1)
    emb("embedding1", (256, 512), 512 + 256)
    cbl("embedding2_cbl", 256, 128, 1)
    emb("embedding2", (128, 256), 256 + 128)
    return out


def det_state(backbone, seed, na=3, nc=80):
    """Deterministic weights exactly as tools/make_golden.py:det_weights fills the reference model."""
    sd = {}
    for i, (k, shp) in enumerate(state_keys(backbone, na, nc)):
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
            continue
        if k.endswith("running_mean"):
            a = np.zeros(shp, np.float32)
        elif k.endswith("running_var"):
            a = np.ones(shp, np.float32)
        elif "bn" in k and k.endswith("weight"):
            a = detrand.uniform(seed + i, shp, 0.5, 1.5)
        elif "bn" in k and k.endswith("bias"):
            a = detrand.uniform(seed + i, shp, -0.2, 0.2)
        elif k.endswith("bias"):
            a = detrand.uniform(seed + i, shp, -0.1, 0.1)
        else:
            fan = shp[1] * shp[2] * shp[3]
            s = math.sqrt(3.0) * math.sqrt(2.0 / fan)
            a = detrand.uniform(seed + i, shp, -s, s)
        sd[k] = torch.from_numpy(a)
    return sd


def forward(sd, x, backbone, training=True, quant=None, record=None, update_running=False):
    """-> (out0, out1, out2).  `quant` optionally rounds activations/weights (e.g. to bf16) to mimic the
    GPU path's storage precision when judging tolerances.  update_running: training mode also updates the running statistics in `sd`
    in place (nn.BatchNorm2d's momentum rule), as a module in train() does - used by the multi-step trajectory test."""
    q = (lambda t: t) if quant is None else quant

    def cbl(name_conv, name_bn, t, stride=1):
        w = q(sd[name_conv + ".weight"])
        k = w.shape[-1]
        z = q(F.conv2d(t, w, stride=stride, padding=(k - 1) // 2))
        if training:
            rm, rv = (sd[name_bn + ".running_mean"], sd[name_bn + ".running_var"]) if update_running else (None, None)
            y = F.batch_norm(z, rm, rv, sd[name_bn + ".weight"], sd[name_bn + ".bias"], True, 0.1, 1e-5)
        else:
            y = F.batch_norm(z, sd[name_bn + ".running_mean"], sd[name_bn + ".running_var"], sd[name_bn + ".weight"],
                             sd[name_bn + ".bias"], False, 0.1, 1e-5)
        y = F.leaky_relu(y, 0.1)
        if record is not None:
            record[name_conv] = (z.detach(), y.detach())
        return y
    x = q(cbl("backbone.conv1", "backbone.bn1", q(x)))
    feats = []
    for li, nb in enumerate(BLOCKS[backbone], 1):
        p = f"backbone.layer{li}"
        x = q(cbl(p + ".ds_conv", p + ".ds_bn", x, 2))
        for b in range(nb):
            r = f"{p}.residual_{b}"
            y = q(cbl(r + ".conv1", r + ".bn1", x))
            y = cbl(r + ".conv2", r + ".bn2", y)
            x = q(y + x)
        feats.append(x)
    x2, x1, x0 = feats[2], feats[3], feats[4]

    def branch(name, t):
        br = None
        for i in range(6):
            t = q(cbl(f"{name}.{i}.conv", f"{name}.{i}.bn", t))
            if i == 4:
                br = t
        out = F.conv2d(t, q(sd[name + ".conv_out.weight"]), sd[name + ".conv_out.bias"])
        return out, br
    out0, b0 = branch("embedding0", x0)
    t = q(cbl("embedding1_cbl.conv", "embedding1_cbl.bn", b0))
    t = torch.cat([F.interpolate(t, scale_factor=2, mode="nearest"), x1], 1)
    out1, b1 = branch("embedding1", t)
    t = q(cbl("embedding2_cbl.conv", "embedding2_cbl.bn", b1))
    t = torch.cat([F.interpolate(t, scale_factor=2, mode="nearest"), x2], 1)
    out2, _ = branch("embedding2", t)
    return out0, out1, out2
