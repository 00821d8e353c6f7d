"""Two processes share the GPU, NO communication: repeated identical steps per plan must agree."""
import os, sys
import torch, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]

def worker(rank):
    sys.path.insert(0, ROOT)
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    from tests.helpers import synth_targets
    eng = YoloV3Engine("darknet_21", 3, 80, device=dev, seed=0)
    tg = [{"bbox": torch.from_numpy(b).to(dev), "category_id": torch.from_numpy(l).to(dev)} for b, l in synth_targets(7 + rank, (3, 2), 80)]
    for i, px in enumerate([64, 128, 192, 128, 192]):
        crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=px).to(dev)
        g = torch.Generator().manual_seed(1000 * rank + i)
        x = torch.randn((2, 3, px, px), generator=g).to(dev)
        runs = []
        for r in range(4):
            eng.train_step(x, tg, crit)
            torch.cuda.synchronize()
            plan = eng._last_plan
            runs.append((eng.flat_g.clone(), [h.clone() for h in plan.heads], {n: rec["a"].buf.clone() for n, rec in plan.layers.items()},
                         {n: (rec["a"].grad.buf.clone() if rec["a"].grad is not None else None) for n, rec in plan.layers.items()}))
        m = float(runs[0][0].abs().max()) + 1e-30
        for r in (1, 2, 3):
            dg = float((runs[r][0] - runs[0][0]).abs().max()) / m
            dh = max(float((a - b).abs().max()) for a, b in zip(runs[r][1], runs[0][1]))
            first = next((n for n in runs[0][2] if not torch.equal(runs[r][2][n], runs[0][2][n])), None)
            order = list(reversed(list(runs[0][3].keys())))
            badg = []
            for n in order:
                a, b = runs[r][3][n], runs[0][3][n]
                if a is None:
                    continue
                e = float((a.float() - b.float()).abs().max()) / (float(b.float().abs().max()) + 1e-30)
                if e > 0.05:
                    badg.append((n, round(e, 3)))
            print(f"rank {rank} px {px} run{r}: grad {dg:.3g} heads {dh:.3g} first differing activation {first}; activation-gradients off by > 5 % (backward order): {badg[:5]} ({len(badg)})", flush=True)

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r,)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]
