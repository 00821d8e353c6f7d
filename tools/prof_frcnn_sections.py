"""Where a Faster R-CNN training step spends its time: per section, host time with and without a device synchronise.
    python tools/prof_frcnn_sections.py [batch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd.optim import FlatSGD
from object_detectors_amd.tvision.frcnn import fasterrcnn_resnet50_fpn
from object_detectors_amd.tvision.roi_heads import fastrcnn_loss
from object_detectors_amd.parallel import step_stream

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
torch.cuda.set_stream(step_stream(dev))
torch.manual_seed(0)
model = fasterrcnn_resnet50_fpn(num_classes=91, device=dev)
eng = model.engine
for sp in eng.specs:
    if sp.bn and sp.bn.endswith(".bn3"):
        eng.buffers[sp.bn + ".weight"].fill_(0.2)
eng.refresh_frozen()
opt = FlatSGD.for_engine(eng, lr=1e-2, momentum=0.9, weight_decay=1e-4)
opt_head = torch.optim.SGD(model.head_parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
g = torch.Generator().manual_seed(0)
imgs = torch.rand((bs, 3, 800, 800), generator=g).to(dev)
targets = []
for _ in range(bs):
    tl = torch.rand((7, 2), generator=g) * 800 * 0.6
    wh = torch.rand((7, 2), generator=g) * 800 * 0.3 + 16
    targets.append({"boxes": torch.cat([tl, tl + wh], 1).to(dev), "labels": torch.randint(1, 91, (7,), generator=g).to(dev)})
model.train()
for _ in range(3):
    opt_head.zero_grad(set_to_none=True); model(imgs, targets); opt.step(); opt_head.step()
torch.cuda.synchronize()
acc = {}
def sec(name, fn):
    t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a = acc.setdefault(name, [0.0, 0.0]); a[0] += t1 - t0; a[1] += t2 - t0
    return r
R = 10
self = model
for _ in range(R):
    opt_head.zero_grad(set_to_none=True)
    n = bs
    image_shapes = [(800, 800)] * n
    self.engine.normalize = True
    out = sec("1 engine forward (body+FPN+RPN head)", lambda: self.engine.forward(imgs, training=True))
    plan = self.engine._last_plan
    rpn_side = sec("2 RPN targets (match, sampler, encode; model: side stream under 1)", lambda: self.rpn_targets.prepare([plan.anchors] * n, targets))
    meta = torch.empty(3 * n, device=dev, dtype=torch.int32)
    boxes, _s = sec("3 proposals (top-k, decode, NMS; no host read)", lambda: self._proposals(out, plan, image_shapes, counts_out=meta[:n]))
    feats = self.engine.feature_maps_nhwc(4)
    proposals, _mi, labels, reg_targets, _pi = sec("4 RoI sampling (roi_match, ONE host read, randperm, roi_sample)",
                                                   lambda: self.roi_targets.select_training_samples_fused(boxes, meta, targets))
    x = sec("5 RoIAlign forward", lambda: self.box_roi_pool.forward_nhwc(feats, proposals, image_shapes))
    cls, reg = sec("6 box head forward (2 FC + predictor)", lambda: self.box_predictor(self.box_head(x)))
    lc, lb = sec("7 fastrcnn_loss", lambda: fastrcnn_loss(cls, reg, [labels], [reg_targets], weights=self.classification_weights, loss_type=self.loss_function_name, class_scale=self.tfidf))
    def rpnl():
        obj = out["cls_logits"].detach().reshape(-1, 1).requires_grad_(True)
        dl = out["bbox_regression"].detach().reshape(-1, 4).requires_grad_(True)
        return obj, dl, self.rpn_targets.losses_prepared(obj, dl, rpn_side)
    obj, dl, rpn_losses = sec("8 RPN losses", rpnl)
    losses = {"loss_classifier": lc, "loss_box_reg": lb}; losses.update(rpn_losses)
    sec("9 torch backward (losses, head, RoIAlign)", lambda: sum(losses.values()).backward())
    sec("10 engine backward", lambda: self.engine.backward(obj.grad, dl.grad, [f.grad for f in feats]))
    sec("11 optimizers", lambda: (opt.step(), opt_head.step()))
tot_h = sum(a[0] for a in acc.values()); tot_s = sum(a[1] for a in acc.values())
print(f"Faster R-CNN R50-FPN training step, bs {bs}, 800 px: per-section time over {R} steps (each section followed by a synchronise)\n")
print("| section | host issue ms | host+device ms |\n|---|---|---|")
for k, (h, s) in acc.items():
    print(f"| {k} | {h / R * 1e3:.2f} | {s / R * 1e3:.2f} |")
print(f"| total | {tot_h / R * 1e3:.2f} | {tot_s / R * 1e3:.2f} |")
