import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import detrand, retina_oracle as ro
from object_detectors_amd.tvision.engine import RetinaNetEngine
dev = torch.device("cuda:0")
import os
from object_detectors_amd._lib import lib
if os.environ.get('TUNE'): lib().mi355det_debug_set(0, int(os.environ['TUNE']))
PX, BS = 128, 2
sd = ro.det_state(7000)
eng = RetinaNetEngine(91, 9, 3, device=dev)
eng.load_reference_state_dict(sd)
x = torch.from_numpy(detrand.uniform(4242, (BS, 3, PX, PX), 0.0, 1.0))
sdg = {k: v.clone() for k, v in sd.items()}
train = [s for s in eng.specs if s.trainable]
for s in train:
    sdg[s.name + ".weight"].requires_grad_(True)
    if s.bias:
        sdg[s.name + ".bias"].requires_grad_(True)
ref = ro.forward(sdg, x)
c1 = torch.from_numpy(detrand.uniform(11, tuple(ref["cls_logits"].shape), -1.0, 1.0)) * 1e-2
c2 = torch.from_numpy(detrand.uniform(12, tuple(ref["bbox_regression"].shape), -1.0, 1.0)) * 1e-2
((ref["cls_logits"] * c1).sum() + (ref["bbox_regression"] * c2).sum()).backward()
eng.forward(x.to(dev), training=True)
eng.backward(c1.to(dev), c2.to(dev))
torch.cuda.synchronize()
got = eng.reference_state_dict(grads=True)
def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))
for s in train:
    for suffix in ([".weight", ".bias"] if s.bias else [".weight"]):
        k = s.name + suffix
        g, r = got[k].cpu(), sdg[k].grad
        print(f"{k:60s} cos {cos(g, r):.5f} ratio {float(g.double().norm() / (r.double().norm() + 1e-30)):.4f}")
print("==== real loss")
import numpy as np
from oracle import tv_oracle as tv
rng = np.random.default_rng(0)
targets, gts = [], []
for i in range(BS):
    m = 3 + i
    tl = rng.uniform(0, PX * 0.5, (m, 2)).astype(np.float32)
    wh = rng.uniform(PX * 0.1, PX * 0.45, (m, 2)).astype(np.float32)
    boxes = np.concatenate([tl, tl + wh], 1)
    labels = rng.integers(1, 91, (m,)).astype(np.int64)
    gts.append((boxes, labels))
    targets.append({"boxes": torch.from_numpy(boxes).to(dev), "labels": torch.from_numpy(labels).to(dev)})
for v in sdg.values():
    v.grad = None
ref = ro.forward(sdg, x)
eng.forward(x.to(dev), training=True)
anchors = eng._last_plan.anchors.cpu().numpy()
cl, rl, mis, (gc, gr) = tv.retinanet_loss(ref["cls_logits"].detach().numpy(), ref["bbox_regression"].detach().numpy(), anchors, gts)
((ref["cls_logits"] * torch.from_numpy(gc)).sum() + (ref["bbox_regression"] * torch.from_numpy(gr)).sum()).backward()
losses = eng.train_step(x.to(dev), targets)
torch.cuda.synchronize()
print("loss", losses.cpu().numpy(), cl, rl)
got = eng.reference_state_dict(grads=True)
for s in train:
    for suffix in ([".weight", ".bias"] if s.bias else [".weight"]):
        k = s.name + suffix
        g, r = got[k].cpu(), sdg[k].grad
        print(f"{k:60s} cos {cos(g, r):.5f} ratio {float(g.double().norm() / (r.double().norm() + 1e-30)):.4f}")
