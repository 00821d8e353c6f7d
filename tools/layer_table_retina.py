#!/usr/bin/env python3
"""Per-layer-family table of the RetinaNet training step (BASELINE configs 3 and 5): every convolution launch of ONE timed step is
bracketed by HIP events on the stream it runs on (forward / data gradient on the step stream, weight gradients on the plan's side stream),
grouped into body stages, FPN, the two towers, cls_logits and bbox_reg, with the algorithmic FLOPs of each launch (2*M*Cout*Cin*k*k from its
conv shape) -> TFLOP/s and fraction of the 2.5 PFLOP/s bf16 MFMA peak.

    python tools/layer_table_retina.py --body resnet50 --classes 91 --batch 16
    python tools/layer_table_retina.py --body resnet101 --classes 1204 --batch 8"""
import argparse
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--px", type=int, default=800)
ap.add_argument("--body", default="resnet50")
ap.add_argument("--classes", type=int, default=91)
ap.add_argument("--dump-us", type=float, default=0.0, help="also list every launch that took at least this many microseconds (stderr), in issue order")
args = ap.parse_args()
from object_detectors_amd._lib import check, lib  # noqa: E402
from object_detectors_amd.optim import FlatSGD  # noqa: E402
from object_detectors_amd.parallel import step_stream  # noqa: E402
from object_detectors_amd.tvision.engine import RetinaNetEngine  # noqa: E402
from object_detectors_amd.yolo.nets.engine import comm_hook  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_stream(step_stream(dev))
eng = RetinaNetEngine(args.classes, 9, 3, device=dev, seed=0, body=args.body)
for sp in eng.specs:
    if sp.bn and sp.bn.endswith(".bn3"):
        eng.buffers[sp.bn + ".weight"].fill_(0.2)
eng.refresh_frozen()
opt = FlatSGD.for_engine(eng, lr=1e-2, momentum=0.9, weight_decay=1e-4)
g = torch.Generator().manual_seed(0)
imgs = torch.rand((args.batch, 3, args.px, args.px), generator=g).to(dev)
targets = []
for _ in range(args.batch):
    tl = torch.rand((7, 2), generator=g) * args.px * 0.6
    wh = torch.rand((7, 2), generator=g) * args.px * 0.3 + 16
    targets.append({"boxes": torch.cat([tl, tl + wh], 1).to(dev), "labels": torch.randint(1, args.classes, (7,), generator=g).to(dev)})
for _ in range(3):
    eng.train_step(imgs, targets)
    opt.step()
torch.cuda.synchronize()
plan = eng._last_plan
L = lib()
kinds = [(L.mi355det_conv_fwd_ex, "fwd"), (L.mi355det_conv_fwd, "fwd"), (L.mi355det_conv_dgrad, "dgrad"), (L.mi355det_conv_dgrad_ws, "dgrad"), (L.mi355det_conv_dgrad_mask, "dgrad"), (L.mi355det_conv_wgrad, "wgrad")]
# layer family of a launch: from the conv shape (the heads' shapes are unique: cin 256, cout = 9*K or 36)
name_of = {}
for rec in plan.ops:
    if rec.get("kind") == "conv":
        nm = rec["name"]
        fam = ("cls_logits" if "cls_logits" in nm else "bbox_reg" if "bbox_reg" in nm else "head towers (cls + reg, 4 x 3x3 each)" if "head." in nm else
               "FPN" if "fpn" in nm else "stem 7x7" if nm.endswith("conv1") and "layer" not in nm else "body " + nm.split("body.")[1].split(".")[0])
        for key in ("shp", "shp_f"):
            name_of[C.addressof(rec[key])] = fam
events = []
detail = []


def run_with_events(calls):
    for fn, a in calls:
        if fn is comm_hook:
            a[0](*a[1:])
            continue
        kind = next((k for f, k in kinds if f is fn), None)
        if kind is None:
            st = fn(*a)
        else:
            shp = a[0]._obj
            sp = a[-1]
            stream = torch.cuda.ExternalStream(sp.value) if isinstance(sp, C.c_void_p) and sp.value else torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            st = fn(*a)
            e1.record(stream)
            fl = 2.0 * shp.n * shp.ho * shp.wo * shp.cout * shp.cin * shp.ksize * shp.ksize
            events.append((kind, name_of.get(C.addressof(shp), f"{shp.cin}->{shp.cout} k{shp.ksize}"), fl, e0, e1, shp.ho))
            detail.append((kind, fn.__name__, name_of.get(C.addressof(shp), "?"), (shp.n, shp.h, shp.w, shp.cin, shp.ho, shp.wo, shp.cout, shp.ksize, shp.stride)))
        if st != 0:
            check(st, fn.__name__)


plan._run = run_with_events
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
eng.train_step(imgs, targets)
opt.step()
t1.record()
torch.cuda.synchronize()
if args.dump_us > 0:
    for (kind, fam, fl, e0, e1, ho), d in zip(events, detail):
        t = e0.elapsed_time(e1) * 1e3
        if t >= args.dump_us:
            print(f"{t:9.1f} us  {d[0]:5s} {d[1]:28s} {d[2]:40s} n,h,w,cin,ho,wo,cout,k,s = {d[3]}", file=sys.stderr)
agg = collections.OrderedDict()
lev = collections.OrderedDict()
for kind, fam, fl, e0, e1, ho in events:
    if fam in ("cls_logits", "bbox_reg"):
        b = lev.setdefault((fam, ho), {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]})[kind]
        b[0] += e0.elapsed_time(e1)
        b[1] += fl
    a = agg.setdefault(fam, {"fwd": [0, 0.0, 0.0], "dgrad": [0, 0.0, 0.0], "wgrad": [0, 0.0, 0.0]})[kind]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += fl
print(f"# RetinaNet {args.body}-FPN, {args.classes} classes, batch {args.batch}, {args.px} px: convolution launches of one training step by layer family\n")
print(f"step with every convolution launch bracketed by events: {t0.elapsed_time(t1):.2f} ms (un-instrumented: see the bench line); TFLOP/s = algorithmic "
      f"2*M*Cout*Cin*k*k / event time; peak = 2500 TFLOP/s dense bf16\n")
print("| layer family | fwd launches | fwd ms | fwd TFLOP/s (frac) | dgrad ms | dgrad TFLOP/s | wgrad ms | wgrad TFLOP/s |\n|---|---|---|---|---|---|---|---|")
tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
for fam, d in agg.items():
    cell = lambda k: (f"{d[k][1]:.3f}", f"{d[k][2] / d[k][1] / 1e9:.0f}" if d[k][1] else "-")
    f, dg, w = cell("fwd"), cell("dgrad"), cell("wgrad")
    frac = f" ({d['fwd'][2] / d['fwd'][1] / 1e9 / 2500:.2f})" if d["fwd"][1] else ""
    print(f"| {fam} | {d['fwd'][0]} | {f[0]} | {f[1]}{frac} | {dg[0]} | {dg[1]} | {w[0]} | {w[1]} |")
    for k in tot:
        tot[k][0] += d[k][1]
        tot[k][1] += d[k][2]
print(f"| all convolutions | | {tot['fwd'][0]:.3f} | {tot['fwd'][1] / tot['fwd'][0] / 1e9:.0f} ({tot['fwd'][1] / tot['fwd'][0] / 1e9 / 2500:.2f}) | {tot['dgrad'][0]:.3f} | "
      f"{tot['dgrad'][1] / max(tot['dgrad'][0], 1e-9) / 1e9:.0f} | {tot['wgrad'][0]:.3f} | {tot['wgrad'][1] / max(tot['wgrad'][0], 1e-9) / 1e9:.0f} |")

print("\n## head output convolutions per pyramid level (ms, TFLOP/s)\n")
print("| layer | map | fwd ms | TF | dgrad ms | TF | wgrad ms | TF |\n|---|---|---|---|---|---|---|---|")
for (fam, ho), d in lev.items():
    c = lambda k: f"{d[k][0]:.3f} | {d[k][1] / d[k][0] / 1e9:.0f}" if d[k][0] else "- | -"
    print(f"| {fam} | {ho}x{ho} | {c('fwd')} | {c('dgrad')} | {c('wgrad')} |")
