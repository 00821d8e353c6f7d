"""Host-side timeline of RetinaNetEngine.train_step (no synchronisation inside): when the host finishes issuing each part, against the
device time of the whole step.   python tools/prof_retina_host.py [--batch 16]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import ops
from object_detectors_amd.optim import FlatSGD
from object_detectors_amd.parallel import step_stream
from object_detectors_amd.tvision.engine import RetinaNetEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_stream(step_stream(dev))
eng = RetinaNetEngine(91, 9, 3, device=dev, seed=0)
for sp in eng.specs:
    if sp.bn and sp.bn.endswith(".bn3"):
        eng.buffers[sp.bn + ".weight"].fill_(0.2)
eng.refresh_frozen()
opt = FlatSGD.for_engine(eng, lr=1e-2, momentum=0.9, weight_decay=1e-4)
g = torch.Generator().manual_seed(0)
imgs = torch.rand((args.batch, 3, 800, 800), generator=g).to(dev)
targets = []
for _ in range(args.batch):
    tl = torch.rand((7, 2), generator=g) * 480
    wh = torch.rand((7, 2), generator=g) * 240 + 16
    targets.append({"boxes": torch.cat([tl, tl + wh], 1).to(dev), "labels": torch.randint(1, 91, (7,), generator=g).to(dev)})
marks = {}
orig = {k: getattr(eng, k) for k in ("forward", "match")}
orig_loss = ops.retina_loss
def wrap(name, fn):
    def f(*a, **k):
        r = fn(*a, **k)
        marks.setdefault(name, []).append(time.perf_counter())
        return r
    return f
eng.forward = wrap("forward issued", orig["forward"])
eng.match = wrap("match issued", orig["match"])
ops.retina_loss = wrap("retina_loss issued", orig_loss)
for _ in range(3):
    eng.train_step(imgs, targets); opt.step()
torch.cuda.synchronize()
marks.clear()
R, t_start, t_end, t_sync = 8, [], [], []
for _ in range(R):
    torch.cuda.synchronize()
    t_start.append(time.perf_counter())
    eng.train_step(imgs, targets); opt.step()
    t_end.append(time.perf_counter())
    torch.cuda.synchronize()
    t_sync.append(time.perf_counter())
avg = lambda xs: sum(xs) / len(xs) * 1e3
print(f"RetinaNet-R50 bs {args.batch}: host timeline of one step (ms after step start, mean of {R}; every step starts with an idle device)")
for k in ("forward issued", "match issued", "retina_loss issued"):
    print(f"  {k:22s} {avg([m - s for m, s in zip(marks[k], t_start)]):7.2f}")
print(f"  {'step issued':22s} {avg([e - s for e, s in zip(t_end, t_start)]):7.2f}")
print(f"  {'device done':22s} {avg([e - s for e, s in zip(t_sync, t_start)]):7.2f}")
