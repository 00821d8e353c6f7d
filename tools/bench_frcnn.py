#!/usr/bin/env python3
"""BASELINE config 4 (one GPU's share): Faster R-CNN ResNet-50-FPN training step, synthetic COCO 800 px.
    python tools/bench_frcnn.py --batch 4 --steps 10"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--px", type=int, default=800)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reuse-rpn-targets", action="store_true", help="MEASUREMENT ONLY: prepare the RPN targets once and reuse them (the "
                    "targets are constant in this bench) - the step time that kernels for RPNTargets.prepare could reach at most; not a valid step")
    ap.add_argument("--tune-record", default="auto", help="tune record to load locked before the plan build: a path, 'none', or 'auto' = "
                    "object_detectors_amd/tune_records/fasterrcnn_resnet50_bs<batch>_<px>.json when it exists (tune.refine_step)")
    ap.add_argument("--refine", default=None, metavar="OUT.json", help="refine the record on the whole training step (tune.refine_step) and write it there")
    ap.add_argument("--refine-budget-s", type=float, default=600.0)
    ap.add_argument("--no-wgrad8", action="store_true", help="A/B: the weight-gradient tuner leaves the 256 x 256 phase-staggered kernel out (debug key 8)")
    args = ap.parse_args()
    if args.no_wgrad8:
        from object_detectors_amd._lib import lib
        lib().mi355det_debug_set(8, 1)
    from object_detectors_amd import tune
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = args.tune_record
    if rec == "auto":
        rec = os.path.join(root, "object_detectors_amd", "tune_records", f"fasterrcnn_resnet50_bs{args.batch}_{args.px}.json")
        if not os.path.exists(rec):
            rec = "none"
    if rec != "none":
        tune.load(rec, replace=False, lock=True)
    from object_detectors_amd.optim import FlatSGD
    from object_detectors_amd.tvision.frcnn import fasterrcnn_resnet50_fpn
    dev = torch.device("cuda:0")
    from object_detectors_amd.parallel import step_stream
    torch.cuda.set_stream(step_stream(dev))      # dependency chain above the side stream (weight gradients), as bench.py
    torch.manual_seed(0)
    model = fasterrcnn_resnet50_fpn(num_classes=91, device=dev)
    eng = model.engine
    for sp in eng.specs:          # stable random-init residual stack (see tools/bench_retina.py)
        if sp.bn and sp.bn.endswith(".bn3"):
            eng.buffers[sp.bn + ".weight"].fill_(0.2)
    eng.refresh_frozen()
    opt = FlatSGD.for_engine(eng, lr=1e-2, momentum=0.9, weight_decay=1e-4)
    opt_head = torch.optim.SGD(model.head_parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand((args.batch, 3, args.px, args.px), generator=g).to(dev)
    targets = []
    for _ in range(args.batch):
        tl = torch.rand((7, 2), generator=g) * args.px * 0.6
        wh = torch.rand((7, 2), generator=g) * args.px * 0.3 + 16
        targets.append({"boxes": torch.cat([tl, tl + wh], 1).to(dev), "labels": torch.randint(1, 91, (7,), generator=g).to(dev)})
    model.train()
    if args.reuse_rpn_targets:
        real, cache = model.rpn_targets.prepare, {}
        model.rpn_targets.prepare = lambda anchors, tg: cache.setdefault("p", real(anchors, tg)) if "p" not in cache else cache["p"]

    def step():
        opt_head.zero_grad(set_to_none=True)
        losses = model(imgs, targets)
        opt.step()
        opt_head.step()
        return losses
    for _ in range(args.warmup):
        l0 = step()
    torch.cuda.synchronize()
    if args.refine:
        os.makedirs(os.path.dirname(os.path.abspath(args.refine)), exist_ok=True)
        # learning rate 0 while refining (see tools/bench_retina.py: a model that diverges on one repeated batch changes the timing)
        lr0 = opt.param_groups[0]["lr"]
        opt.param_groups[0]["lr"] = 0.0
        for gq in opt_head.param_groups:
            gq["lr"] = 0.0
        a, b, kept = tune.refine_step(step, rounds=2, steps=6, min_gain_us=30.0, budget_s=args.refine_budget_s,
                                      log=lambda m: print(m, file=sys.stderr, flush=True), checkpoint=args.refine)
        opt.param_groups[0]["lr"] = lr0
        for gq in opt_head.param_groups:
            gq["lr"] = lr0
        tune.save(args.refine)
        print(f"refined: {a:.0f} -> {b:.0f} us per step, {kept} entries changed", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l1 = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    model.eval()
    with torch.no_grad():
        model(imgs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            det = model(imgs)
        torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"bench": "fasterrcnn_resnet50_fpn" + ("_REUSED_RPN_TARGETS_not_a_valid_step" if args.reuse_rpn_targets else ""), "batch": args.batch, "px": args.px, "train_images_per_s": round(args.batch / dt, 2),
                      "train_ms_per_step": round(dt * 1e3, 2), "eval_images_per_s": round(args.batch / de, 2), "eval_ms_per_batch": round(de * 1e3, 2),
                      "losses_first": {k: round(float(v), 4) for k, v in l0.items()}, "losses_last": {k: round(float(v), 4) for k, v in l1.items()},
                      "detections_img0": int(det[0]["boxes"].shape[0]),
                      "tune_record": (os.path.relpath(rec, root) if rec != "none" else None)}))


if __name__ == "__main__":
    main()
