#!/usr/bin/env python3
"""RetinaNet inference (BASELINE configs 3 / 5 in eval mode): network forward + RetinaNet.postprocess_detections (per-level score threshold +
top-1000, decode, clip, per-class batched NMS, first 300; retinanet.py:414-472), synthetic 800 px batch with random-init weights (the
cls_logits bias carries the reference's 0.01 prior, retinanet.py:90-91, so only the tail of the random logits passes the 0.05 threshold).
    python tools/bench_retina_infer.py [--body resnet101 --classes 1204 --batch 8]"""
import argparse
import json
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
    ap.add_argument("--body", default="resnet50")
    ap.add_argument("--classes", type=int, default=91)
    ap.add_argument("--prior", type=float, default=0.05, help="probability put into the cls_logits bias: 0.05 = the score threshold, so about half of the "
                    "random logits pass it and every level delivers its full top-1000 (worst case for the NMS); 0.01 = the reference's initialisation")
    args = ap.parse_args()
    from object_detectors_amd.tvision.retinanet import RetinaNet
    dev = torch.device("cuda:0")
    model = RetinaNet(args.classes, 3, device=dev, body=args.body)
    eng = model.engine
    for sp in eng.specs:
        if sp.bn and sp.bn.endswith(".bn3"):
            eng.buffers[sp.bn + ".weight"].fill_(0.2)
    eng.refresh_frozen()
    import math
    eng.params["head.classification_head.cls_logits.bias"].fill_(math.log(args.prior / (1.0 - args.prior)))
    model.eval()
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand((args.batch, 3, args.px, args.px), generator=g).to(eng.device)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, out

    with torch.no_grad():
        t_net, _ = timed(lambda: eng.forward(imgs, training=False))
        t_all, det = timed(lambda: model(imgs))
    nd = [int(d["boxes"].shape[0]) for d in det]
    print(json.dumps({"bench": f"retinanet_{args.body}_fpn_inference_{args.classes}cls", "batch": args.batch, "px": args.px,
                      "network_ms": round(t_net * 1e3, 3), "end_to_end_ms": round(t_all * 1e3, 3), "postprocess_ms": round((t_all - t_net) * 1e3, 3),
                      "end_to_end_images_per_s": round(args.batch / t_all, 1), "cls_bias_prior": args.prior, "detections_per_image": nd[:4]}))


if __name__ == "__main__":
    main()
