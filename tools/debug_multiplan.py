"""Repro hunt: two identical training steps on one plan must give the same gradients (up to the BN-backward atomics, ~6e-4)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
from object_detectors_amd.yolo.nets.engine import YoloV3Engine
from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
from tests.helpers import synth_targets
eng = YoloV3Engine("darknet_21", 3, 80, device=dev, seed=0)
tg = [{"bbox": torch.from_numpy(b).to(dev), "category_id": torch.from_numpy(l).to(dev)} for b, l in synth_targets(7, (3, 2), 80)]
sizes = [int(v) for v in sys.argv[1:]] or [64, 128, 192, 128, 192, 192]
for i, px in enumerate(sizes):
    crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=px).to(dev)
    g = torch.Generator().manual_seed(i)
    x = torch.randn((2, 3, px, px), generator=g).to(dev)
    runs = []
    for r in range(3):
        out12 = eng.train_step(x, tg, crit)
        torch.cuda.synchronize()
        plan = eng._last_plan
        heads = [h.clone() for h in plan.heads]
        acts = {n: rec["a"].buf.clone() for n, rec in plan.layers.items()}
        runs.append((eng.flat_g.clone(), heads, acts, out12.clone()))
    m = float(runs[0][0].abs().max()) + 1e-30
    msg = f"px {px}:"
    for r in (1, 2):
        dg = float((runs[r][0] - runs[0][0]).abs().max()) / m
        dh = max(float((a - b).abs().max()) for a, b in zip(runs[r][1], runs[0][1]))
        first = next((n for n in runs[0][2] if not torch.equal(runs[r][2][n], runs[0][2][n])), None)
        msg += f"  run{r}: grad {dg:.3g} heads {dh:.3g} first differing activation {first}"
    print(msg, flush=True)
    if px >= 192:
        rows = []
        for name, o, n, _s in eng.param_order:
            a, b = runs[0][0][o:o + n], runs[1][0][o:o + n]
            rows.append((name, float((a - b).abs().max()) / (float(a.abs().max()) + 1e-30)))
        print("   per tensor (forward order):", [(n, f"{e:.2g}") for n, e in rows if e > 1e-3][:200], flush=True)
