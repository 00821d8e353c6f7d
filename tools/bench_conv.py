"""Per-layer conv micro-benchmark on the GPU box: fwd / dgrad / wgrad TFLOP/s for the YOLOv3@640 bs-32 shapes.
    python tools/bench_conv.py [tune values ...]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd._lib import lib
dev = torch.device('cuda:0')
SHAPES = [  # n, h, w, cin, cout, k, s, launches per YOLOv3 step
    (32, 80, 80, 128, 256, 3, 1, 11), (32, 40, 40, 256, 512, 3, 1, 11), (32, 20, 20, 512, 1024, 3, 1, 7),
    (32, 40, 40, 512, 256, 1, 1, 10), (32, 80, 80, 256, 128, 1, 1, 10), (32, 640, 640, 32, 64, 3, 2, 1),
    (32, 160, 160, 64, 128, 3, 1, 2), (32, 320, 320, 32, 64, 3, 1, 1), (32, 20, 20, 1024, 512, 1, 1, 7),
    (32, 320, 320, 64, 128, 3, 2, 1), (32, 160, 160, 128, 256, 3, 2, 1), (32, 160, 160, 128, 64, 1, 1, 2),
    (32, 320, 320, 64, 32, 1, 1, 1), (32, 40, 40, 512, 1024, 3, 2, 1), (32, 80, 80, 256, 512, 3, 2, 1),
    (32, 80, 80, 384, 128, 1, 1, 1), (32, 40, 40, 768, 256, 1, 1, 1), (32, 40, 40, 256, 128, 1, 1, 1), (32, 20, 20, 512, 256, 1, 1, 1),
]
tunes = [int(v) for v in sys.argv[1:]] or [0]
which = 'fdw'
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tot = [0.0, 0.0, 0.0]
for (n, h, w, cin, cout, k, s, cnt) in SHAPES:
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    x = (torch.randn(n, h, w, cin, device=dev)).bfloat16()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wf, wd = ops.pack_weights(shape, wt)
    y = torch.empty(n, shape.ho, shape.wo, cout, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(n, shape.ho, shape.wo, cout, device=dev).bfloat16()
    dx = torch.empty_like(x)
    dw = torch.zeros(cout, k * k * cin, device=dev)
    rows = ops.conv_stats_rows(shape)
    stats = torch.zeros(rows + 64, 2, ops.cout_pad_of(cout), device=dev)
    fl = 2.0 * n * shape.ho * shape.wo * cout * cin * k * k
    msg = f"{cin:4d}->{cout:4d} k{k} s{s} @{shape.ho:3d}: "
    # same tuning the engine does at plan build: first launch of a shape under autotune mode times the tile candidates
    import ctypes as C
    lib().mi355det_conv_autotune_mode(1)
    ops.conv_fwd(shape, x, wf, y, stats=stats); ops.conv_dgrad(shape, dy, wd, dx)
    lib().mi355det_conv_autotune_mode(0)
    ws = torch.empty(lib().mi355det_conv_wgrad_workspace(C.byref(shape)), device=dev, dtype=torch.uint8)
    lib().mi355det_conv_wgrad_autotune(C.byref(shape), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), None)
    for t in tunes:
        lib().mi355det_debug_set(0, t)
        us = timeit(lambda: ops.conv_fwd(shape, x, wf, y, stats=stats))
        msg += f" fwd[t{t}] {us:7.1f}us {fl / us / 1e6:6.0f}TF |"
        if t == tunes[0]: tot[0] += cnt * us
    lib().mi355det_debug_set(0, 0)
    us = timeit(lambda: ops.conv_dgrad(shape, dy, wd, dx)); tot[1] += cnt * us; msg += f" dgrad {us:7.1f}us {fl / us / 1e6:6.0f}TF |"
    us = timeit(lambda: ops.conv_wgrad(shape, x, dy, dw, workspace=ws)); tot[2] += cnt * us; msg += f" wgrad {us:7.1f}us {fl / us / 1e6:6.0f}TF"
    # floors: dense bf16 MFMA peak 2.5 PFLOP/s; algorithmic HBM bytes (each operand once, bf16 tensors, fp32 dW) at 8 TB/s
    px_in, px_out = n * h * w, n * shape.ho * shape.wo
    by = {"f": 2 * (px_in * cin + px_out * cout), "d": 2 * (px_in * cin + px_out * cout), "w": 2 * (px_in * cin + px_out * cout) + 4 * cout * cin * k * k}
    msg += "  || floor us (mfma, hbm): " + f"{fl / 2.5e9:5.1f} " + " ".join(f"{b / 8e6:5.1f}" for b in by.values())
    print(f"x{cnt:2d} " + msg, flush=True)
print(f"sum over the step's launches (ms): fwd {tot[0] / 1e3:.2f}  dgrad {tot[1] / 1e3:.2f}  wgrad {tot[2] / 1e3:.2f}")
