"""Device time of mi355det_rpn_proposals (Faster R-CNN training shapes: 4 x 159882 anchors, 5 levels, 2000 / 2000) and of the composed
route; under `rocprofv3 --kernel-trace --stats` the per-kernel split.     python tools/bench_proposals.py [--batch 4] [--composed]"""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import ops
from object_detectors_amd.tvision.postprocess import rpn_filter_proposals, rpn_proposals_fused

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--pre", type=int, default=2000)
ap.add_argument("--composed", action="store_true")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, levels = args.batch, [200 * 200 * 3, 100 * 100 * 3, 50 * 50 * 3, 25 * 25 * 3, 13 * 13 * 3]
A = sum(levels)
g = torch.Generator(device=dev).manual_seed(0)
obj = torch.randn((N, A), device=dev, generator=g)
ctr = torch.rand((A, 2), device=dev, generator=g) * 800
wh = torch.rand((A, 2), device=dev, generator=g) * 200 + 4
anchors = torch.cat([ctr - wh / 2, ctr + wh / 2], -1)
deltas = torch.randn((N, A, 4), device=dev, generator=g) * 0.1
shapes = [(800, 800)] * N
clip = math.log(1000.0 / 16)

def run():
    if args.composed:
        props = ops.box_decode(deltas.reshape(-1, 4), anchors.repeat(N, 1), (1.0, 1.0, 1.0, 1.0), clip).reshape(N, -1, 4)
        return rpn_filter_proposals(props, obj, shapes, levels, args.pre, args.pre)
    return rpn_proposals_fused(deltas, obj, anchors, shapes, levels, args.pre, args.pre, xform_clip=clip)

for _ in range(3):
    out = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
t0 = time.perf_counter()
e0.record()
for _ in range(args.reps):
    out = run()
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.reps * 1e3
print(f"{'composed' if args.composed else 'fused'} bs {N} pre {args.pre}: {e0.elapsed_time(e1) / args.reps:.3f} ms per call (events), {wall:.3f} ms wall; "
      f"kept {[int(b.shape[0]) for b in out[0]]}")
