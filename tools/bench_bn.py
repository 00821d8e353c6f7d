"""Isolated timing of the three BatchNorm+LeakyReLU passes on the YOLOv3@640 bs-32 activation shapes (TB/s of algorithmic bytes:
forward 4 B, backward reduce 4 B, backward apply 6 B per element).   python tools/bench_bn.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd._lib import check, lib, ptr, stream_ptr
dev = torch.device('cuda:0')
SHAPES = [(32, 640), (64, 320), (32, 320), (128, 160), (64, 160), (256, 80), (128, 80), (512, 40), (256, 40), (1024, 20), (512, 20)]   # channels, map
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for c, hw in SHAPES:
    pixels = 32 * hw * hw
    z = torch.randn(pixels, c, device=dev).bfloat16()
    g = torch.randn(pixels, c, device=dev).bfloat16()
    out = torch.empty_like(z)
    ss = torch.cat([torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)]).to(dev)
    sums = torch.zeros(2 * c, device=dev)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    L = lib()
    f = timeit(lambda: check(L.mi355det_bn_act_fwd(ptr(z), c, ptr(ss), c, pixels, 0.1, None, 0, ptr(out), c, stream_ptr())))
    r = timeit(lambda: check(L.mi355det_bn_act_bwd_reduce(ptr(g), c, None, 0, ptr(z), c, ptr(ss), c, pixels, 0.1, ptr(sums), stream_ptr())))
    nb = L.mi355det_bn_act_bwd_reduce_workspace(c, pixels)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    rd = timeit(lambda: check(L.mi355det_bn_act_bwd_reduce_det(ptr(g), c, None, 0, ptr(z), c, ptr(ss), c, pixels, 0.1, ptr(sums), ptr(ws), nb, stream_ptr())))
    a = timeit(lambda: check(L.mi355det_bn_act_bwd_apply(ptr(g), c, None, 0, ptr(z), c, ptr(ss), ptr(sums), None, c, pixels, 0.1, ptr(out), c, ptr(dg), ptr(db), stream_ptr())))
    e = pixels * c
    print(f"c={c:5d} @{hw:3d}  {e * 2 / 1e6:7.1f} MB/tensor | fwd {f:7.1f} us {e * 4 / f / 1e6:5.2f} TB/s | reduce (atomics) {r:7.1f} us {e * 4 / r / 1e6:5.2f} TB/s | reduce (fixed order) {rd:7.1f} us {e * 4 / rd / 1e6:5.2f} TB/s | apply {a:7.1f} us {e * 6 / a / 1e6:5.2f} TB/s", flush=True)
