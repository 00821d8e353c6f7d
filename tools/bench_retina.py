#!/usr/bin/env python3
"""BASELINE config 3: RetinaNet ResNet-50-FPN training step (fwd + matcher + focal/L1 loss + bwd + SGD), synthetic COCO 800 px.
    python tools/bench_retina.py --batch 16 --px 800 --steps 10"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--px", type=int, default=800)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-fwd-only", action="store_true", help="skip the trailing forward-only timing (profiling passes: the trace then ends with whole steps)")
    ap.add_argument("--body", default="resnet50")
    ap.add_argument("--classes", type=int, default=91)
    ap.add_argument("--tune-record", default="auto", help="tune record to load locked before the plan build: a path, 'none', or 'auto' = "
                    "object_detectors_amd/tune_records/retinanet_<body>_<classes>cls_bs<batch>_<px>.json when it exists (tune.refine_step)")
    ap.add_argument("--refine", default=None, metavar="OUT.json", help="refine the record on the whole step (tune.refine_step) and write it there")
    ap.add_argument("--refine-budget-s", type=float, default=600.0)
    ap.add_argument("--refine-min-gain-us", type=float, default=40.0, help="a change is kept only if the step gets faster by more than this (long steps are noisier: "
                    "the first R101-LVIS record, refined with 40 us on a 40 ms step, was 1 %% SLOWER than no record on another box)")
    ap.add_argument("--refine-steps", type=int, default=4)
    ap.add_argument("--no-wgrad8", action="store_true", help="A/B: the weight-gradient tuner leaves the 256 x 256 phase-staggered kernel out (debug key 8)")
    args = ap.parse_args()
    if args.no_wgrad8:
        from object_detectors_amd._lib import lib
        lib().mi355det_debug_set(8, 1)
    from object_detectors_amd import tune
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = args.tune_record
    if rec == "auto":
        rec = os.path.join(root, "object_detectors_amd", "tune_records", f"retinanet_{args.body}_{args.classes}cls_bs{args.batch}_{args.px}.json")
        if not os.path.exists(rec):
            rec = "none"
    if rec != "none":
        tune.load(rec, replace=False, lock=True)
    from object_detectors_amd.optim import FlatSGD
    from object_detectors_amd.tvision.engine import RetinaNetEngine
    dev = torch.device("cuda:0")
    from object_detectors_amd.parallel import step_stream
    torch.cuda.set_stream(step_stream(dev))      # dependency chain above the side stream (weight gradients), as bench.py
    eng = RetinaNetEngine(args.classes, 9, 3, device=dev, seed=0, body=args.body)
    # random-init residual stacks with FROZEN BatchNorm have no normalisation at all: damp the last BN of every bottleneck
    # (as zero-init-residual / pretrained weights do) so that 33 blocks of ResNet-101 stay finite in bf16
    for sp in eng.specs:
        if sp.bn and sp.bn.endswith(".bn3"):
            eng.buffers[sp.bn + ".weight"].fill_(0.2)
    eng.refresh_frozen()
    opt = FlatSGD.for_engine(eng, lr=1e-2, momentum=0.9, weight_decay=1e-4)      # detection/train.py default lr 0.02 at 8 GPUs x 2 images
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand((args.batch, 3, args.px, args.px), generator=g).to(dev)
    targets = []
    for _ in range(args.batch):
        tl = torch.rand((7, 2), generator=g) * args.px * 0.6
        wh = torch.rand((7, 2), generator=g) * args.px * 0.3 + 16
        targets.append({"boxes": torch.cat([tl, tl + wh], 1).to(dev), "labels": torch.randint(1, args.classes, (7,), generator=g).to(dev)})

    def step():
        losses = eng.train_step(imgs, targets)
        opt.step()
        return losses
    t0 = time.perf_counter()
    for _ in range(args.warmup):
        l0 = step()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    if args.refine:
        os.makedirs(os.path.dirname(os.path.abspath(args.refine)), exist_ok=True)
        # learning rate 0 while refining: thousands of SGD steps on ONE synthetic batch diverge (lr 0.01: NaN losses after a few hundred steps),
        # and a step on NaN operands is 7 % FASTER (33.3 against 36 ms: round 4's first two R101 records were refined on a diverged model)
        lr0, opt.param_groups[0]["lr"] = opt.param_groups[0]["lr"], 0.0
        a, b, kept = tune.refine_step(step, rounds=2, steps=args.refine_steps, min_gain_us=args.refine_min_gain_us, budget_s=args.refine_budget_s,
                                      log=lambda m: print(m, file=sys.stderr, flush=True), checkpoint=args.refine)
        opt.param_groups[0]["lr"] = lr0
        if not all(math.isfinite(float(v)) for v in step()):
            raise SystemExit("the model diverged during the refinement: its timings are not those of a training step")
        tune.save(args.refine)
        print(f"refined: {a:.0f} -> {b:.0f} us per step, {kept} entries changed", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l1 = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # forward-only timing
    t0 = time.perf_counter()
    for _ in range(0 if args.no_fwd_only else args.steps):
        eng.forward(imgs, training=True)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"bench": f"retinanet_{args.body}_fpn_train_step_{args.classes}cls", "batch": args.batch, "px": args.px, "images_per_s": round(args.batch / dt, 2),
                      "ms_per_step": round(dt * 1e3, 3), "fwd_ms": round(df * 1e3, 3), "plan_build_s": round(t_build, 1),
                      "loss_first": [round(float(v), 4) for v in l0], "loss_last": [round(float(v), 4) for v in l1],
                      "tune_record": (os.path.relpath(rec, root) if rec != "none" else None), "trainable_params": int(eng.flat_w.numel()), "anchors_per_image": int(eng._last_plan.rows)}))


if __name__ == "__main__":
    main()
