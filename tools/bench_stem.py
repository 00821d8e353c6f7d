"""Isolated timing of the four stem recompute kernels (csrc/stem_kernels.hip) at the BASELINE size, with their HBM floors.
    python tools/bench_stem.py [batch] [px]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd._lib import check, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
px = int(sys.argv[2]) if len(sys.argv) > 2 else 640
dev = torch.device("cuda:0")
L = lib()
vp = lambda t: C.c_void_p(t.data_ptr())
img = torch.randn((n, 3, px, px), device=dev)
wp = torch.zeros((32, 32), dtype=torch.bfloat16, device=dev)
wp[:, :27] = (torch.randn((32, 27), device=dev) * 0.27).bfloat16()
rows = L.mi355det_stem_rows(n, px, px)
part = torch.zeros((rows + 64, 2, 32), device=dev)
ss = torch.zeros(128, device=dev)
gam, bet = torch.ones(32, device=dev), torch.zeros(32, device=dev)
rm, rv = torch.zeros(32, device=dev), torch.ones(32, device=dev)
a = torch.empty((n, px, px, 32), dtype=torch.bfloat16, device=dev)
da = (torch.randn((n, px, px, 32), device=dev) * 0.05).bfloat16()
sums = torch.zeros(64, device=dev)
slab = torch.zeros((rows, 1024), device=dev)
dw, dg, db = torch.zeros((32, 32), device=dev), torch.zeros(32, device=dev), torch.zeros(32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
pix = n * px * px
calls = {
    "stem_fwd_stats": (lambda: L.mi355det_stem_fwd_stats(vp(img), vp(wp), vp(part), n, px, px, st), pix * 12),
    "bn_finalize": (lambda: L.mi355det_bn_finalize(vp(part), rows, 32, 32, pix, vp(gam), vp(bet), 1e-5, 0.1, vp(rm), vp(rv), vp(ss), st), 0),
    "stem_fwd_apply": (lambda: L.mi355det_stem_fwd_apply(vp(img), vp(wp), vp(ss), 0.1, vp(a), 32, n, px, px, st), pix * (12 + 64)),
    "stem_bwd_reduce": (lambda: L.mi355det_stem_bwd_reduce(vp(img), vp(wp), vp(ss), 0.1, vp(da), 32, vp(part), n, px, px, st), pix * (12 + 64)),
    "bn_bwd_sum_partials": (lambda: L.mi355det_bn_bwd_sum_partials(vp(part), rows, 32, 32, vp(sums), st), 0),
    "stem_bwd_apply_wgrad": (lambda: L.mi355det_stem_bwd_apply_wgrad(vp(img), vp(wp), vp(ss), vp(sums), 0.1, vp(da), 32, vp(slab), vp(dw), vp(dg),
                                                                    vp(db), n, px, px, st), pix * (12 + 64)),
}
slab4, ag4 = torch.zeros((rows, 2048), device=dev), torch.zeros(2048, device=dev)
calls["stem_bwd_fused (one pass: replaces reduce + apply_wgrad)"] = (lambda: L.mi355det_stem_bwd_fused(vp(img), vp(wp), vp(ss), 0.1, vp(da), 32, vp(slab4), vp(ag4),
                                                                                                   vp(sums), n, px, px, st), pix * (12 + 64))
calls["stem_bwd_finish"] = (lambda: L.mi355det_stem_bwd_finish(vp(wp), vp(ss), vp(ag4), vp(sums), pix, vp(dw), vp(dg), vp(db), st), 0)
from object_detectors_amd import ops  # noqa: E402
shp1 = ops.conv_shape(n, px, px, 32, 64, 3, 2)
wf1, _ = ops.pack_weights(shp1, torch.randn(64, 32, 3, 3, device=dev) * 0.08)
rows1 = L.mi355det_stem_l1_rows(n, px, px)
z1 = torch.empty((n, px // 2, px // 2, 64), dtype=torch.bfloat16, device=dev)
stats1 = torch.zeros((rows1 + 64, 2, 64), device=dev)
calls["stem_l1_fwd (a0 side output)"] = (lambda: L.mi355det_stem_l1_fwd(vp(img), vp(wp), vp(ss), 0.1, vp(wf1), vp(a), 32, vp(z1), 64, vp(stats1), n, px, px, st),
                                          pix * (12 + 64 + 32))
calls["stem_l1_fwd (no side output)"] = (lambda: L.mi355det_stem_l1_fwd(vp(img), vp(wp), vp(ss), 0.1, vp(wf1), None, 0, vp(z1), 64, vp(stats1), n, px, px, st),
                                          pix * (12 + 32))
for name, (fn, nbytes) in calls.items():
    check(fn(), name)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    floor = nbytes / 6.3e12 * 1e6
    print(f"{name:24s} {best:8.1f} us   algorithmic {nbytes / 1e6:8.1f} MB   floor@6.3TB/s {floor:6.1f} us   {nbytes / best / 1e6:6.2f} TB/s", flush=True)
