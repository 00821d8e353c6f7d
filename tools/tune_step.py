#!/usr/bin/env python3
"""Step-level refinement of a tune record (YOLOv3-D53 training step of bench.py).

The plan build times every candidate ALONE (a kernel back to back with itself, warm caches, the whole chip); inside the step a data gradient
shares the chip with a weight gradient on the other stream, every forward convolution starts behind a streaming BatchNorm pass, and the
clock depends on what ran before.  This tool takes the record the plan build produced and does coordinate descent on the WHOLE STEP: for one
entry at a time (a shape: all layers of that shape move together) it tries the other legal values, measures the step, and keeps a value only
if the step got faster by more than the noise, confirmed by a second measurement.  The choices are looked up in the library at launch time,
so a trial is an import of a modified record + a few steps: no plan rebuild.

    python tools/tune_step.py --out object_detectors_amd/tune_records/yolov3_d53_bs32_640.json [--rounds 2] [--storage bf16]

The result is an ordinary tune record (`MI355DET_TUNE_LOAD`, `bench.py --tune-record`): reproducible kernels AND the step-level choice."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (synthetic batch, anchors)
from object_detectors_amd import tune  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--start", default=None, help="record to start from (default: what the plan build times on this box)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--px", type=int, default=640)
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--storage", default="bf16")
    ap.add_argument("--min-gain-us", type=float, default=40.0)
    ap.add_argument("--budget-s", type=float, default=900.0)
    ap.add_argument("--skip", type=int, default=0, help="skip the first N entries of the sweep (resume a sweep that ran out of time)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from object_detectors_amd.optim import FlatSGD
    from object_detectors_amd.parallel import step_stream
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    torch.cuda.set_stream(step_stream(dev))
    if args.start:
        tune.load(args.start, replace=True, lock=True)
    eng = YoloV3Engine("darknet_53", 3, 80, device=dev, seed=0, storage=args.storage)
    crit = YOLOForw(anchors=bench.ANCHORS, num_classes=80, img_size=args.px).to(dev)
    imgs, targets = bench.synth_batch(args.batch, args.px, 0, dev)
    opt = FlatSGD.for_engine(eng, lr=0.0, momentum=0.9, weight_decay=5e-4)      # lr 0: the model must not drift (or diverge) over the thousands of steps of a sweep
    S = 1024.0 if args.storage == "fp16" else 1.0

    def step():
        eng.train_step(imgs, targets, crit, grad_scale=S)
        opt.step(grad_scale=1.0 / S)

    for _ in range(3):
        step()                                   # plan build + the isolated tuning pass (shapes the start record does not cover)
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    lines = []

    def log(msg):
        lines.append(msg)
        print(msg, flush=True)

    start, final, kept = tune.refine_step(step, storage=args.storage, rounds=args.rounds, steps=args.steps, min_gain_us=args.min_gain_us,
                                          budget_s=args.budget_s, skip=args.skip, log=log, checkpoint=args.out)
    tune.save(args.out)
    with open(os.path.splitext(args.out)[0] + ".log", "w") as f:
        f.write("\n".join(l for l in lines if not l.startswith("  [")) + "\n")
    print(json.dumps({"start_us": round(start), "final_us": round(final), "changed": kept, "record": args.out}))


if __name__ == "__main__":
    main()
