"""A/B of the pixel-tile height of igemm8_kernel (tile configurations 40 / 44 / 45 = 256 / 224 / 208 pixels; 46 = 192 existed in round 4 only) on the three big
3x3 families and the stride-2 / 1x1 layers that run it (YOLOv3 @640 bs 32), forward and data gradient, interleaved rounds in one process;
0 = the autotuned choice among ALL candidates.      python tools/bench_tiles.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd._lib import lib
dev = torch.device('cuda:0')
SHAPES = [(32, 80, 80, 128, 256, 3, 1), (32, 40, 40, 256, 512, 3, 1), (32, 20, 20, 512, 1024, 3, 1), (32, 40, 40, 512, 256, 1, 1),
          (32, 20, 20, 1024, 512, 1, 1), (32, 80, 80, 256, 512, 3, 2), (32, 40, 40, 512, 1024, 3, 2)]
cfgs = [int(v) for v in sys.argv[1:]] or [0, 40, 44, 45]
def timeit(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (n, h, w, cin, cout, k, s) in SHAPES:
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    x = torch.randn(n, h, w, cin, device=dev).bfloat16()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wf, wd = ops.pack_weights(shape, wt)
    y = torch.empty(n, shape.ho, shape.wo, cout, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(n, shape.ho, shape.wo, cout, device=dev).bfloat16()
    dx = torch.empty_like(x)
    stats = torch.zeros(ops.conv_stats_rows(shape) + 64, 2, ops.cout_pad_of(cout), device=dev)
    fl = 2.0 * n * shape.ho * shape.wo * cout * cin * k * k
    lib().mi355det_conv_autotune_mode(1)
    ops.conv_fwd(shape, x, wf, y, stats=stats); ops.conv_dgrad(shape, dy, wd, dx)
    lib().mi355det_conv_autotune_mode(0)
    best = {}
    for rnd in range(3):
        for c in cfgs:
            lib().mi355det_debug_set(0, c)
            f = lambda: ops.conv_fwd(shape, x, wf, y, stats=stats)
            d = lambda: ops.conv_dgrad(shape, dy, wd, dx)
            f(); d(); torch.cuda.synchronize()
            tf, td = timeit(f), timeit(d)
            b = best.setdefault(c, [1e9, 1e9]); b[0] = min(b[0], tf); b[1] = min(b[1], td)
    lib().mi355det_debug_set(0, 0)
    M = n * shape.ho * shape.wo
    msg = f"{cin:4d}->{cout:4d} k{k} s{s} @{shape.ho:3d} (tiles 256/224/208/192: {-(-M // 256) * (cout // 256)}/{-(-M // 224) * (cout // 256)}/{-(-M // 208) * (cout // 256)}/{-(-M // 192) * (cout // 256)}):"
    for c in cfgs:
        msg += f"  [cfg{c:2d}] fwd {best[c][0]:6.1f}us {fl / best[c][0] / 1e6:5.0f}TF dgrad {best[c][1]:6.1f}us {fl / best[c][1] / 1e6:5.0f}TF |"
    print(msg, flush=True)
