"""Isolated weight-gradient timing on the YOLOv3 @640 bs-32 shapes, with the ablation builds of wgrad_kernel (mi355det_debug_set(6, v):
1 = no X-tile LDS-DMA, 2 = no dY-tile LDS-DMA, 4 = no MFMAs; timing only).      python tools/bench_wgrad.py [ablate ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd._lib import lib
import ctypes as C
dev = torch.device('cuda:0')
SHAPES = [(32, 80, 80, 128, 256, 3, 1), (32, 40, 40, 256, 512, 3, 1), (32, 20, 20, 512, 1024, 3, 1), (32, 40, 40, 512, 256, 1, 1),
          (32, 80, 80, 256, 128, 1, 1), (32, 160, 160, 64, 128, 3, 1), (32, 320, 320, 32, 64, 3, 1)]
if "retina" in sys.argv[1:]:      # RetinaNet-R101-LVIS bs 8 @800: cls_logits (10 836 channels, pitch 10 880) and a tower convolution on the 100 x 100 level
    SHAPES = [(8, 100, 100, 256, 10836, 3, 1), (8, 100, 100, 256, 256, 3, 1), (16, 100, 100, 256, 819, 3, 1),
              (8, 50, 50, 256, 10836, 3, 1), (8, 25, 25, 256, 10836, 3, 1), (8, 50, 50, 256, 256, 3, 1)]
    sys.argv.remove("retina")
abl = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4]
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
L = lib()
for (n, h, w, cin, cout, k, s) in SHAPES:
    ld = (cout + 63) // 64 * 64
    shape = ops.conv_shape(n, h, w, cin, cout, k, s, out_ld=ld)
    x = torch.randn(n, h, w, cin, device=dev).bfloat16()
    dy = torch.randn(n, shape.ho, shape.wo, ld, device=dev).bfloat16()
    dw = torch.zeros(cout, k * k * cin, device=dev)
    need = L.mi355det_conv_wgrad_workspace(C.byref(shape))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    sp = L.mi355det_conv_wgrad_autotune(C.byref(shape), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), None)
    fl = 2.0 * n * shape.ho * shape.wo * cout * cin * k * k
    msg = f"{cin:4d}->{cout:4d} k{k} s{s} @{shape.ho:3d} splits {sp:3d}:"
    for a in abl:
        L.mi355det_debug_set(6, a)
        t = min(timeit(lambda: ops.conv_wgrad(shape, x, dy, dw, workspace=ws)) for _ in range(3))
        msg += f"  [abl {a}] {t:6.1f}us {fl / t / 1e6:5.0f}TF |"
    L.mi355det_debug_set(6, 0)
    print(msg, flush=True)
    # the 256 x 256 phase-staggered kernel (forced through debug key 7: split + 65536) against the 128 x 128 kernel at the same split counts
    if cout >= 256 and k * k * cin >= 256 and shape.wo >= 4:
        t8 = ((cout + 255) // 256) * ((k * k * cin + 255) // 256)
        msg = "      form8 / form128 by splits:"
        M = n * shape.ho * shape.wo
        def valid(sp):      # csrc/wgrad_kernels.hip: split_valid (chunks are whole 64-pixel k-steps)
            chunk = ((M + sp - 1) // sp + 63) // 64 * 64
            return (M + chunk - 1) // chunk == sp
        def near(sp):
            while sp > 1 and not valid(sp):
                sp -= 1
            return max(1, sp)
        for spc in sorted({near(128 // t8), near(256 // t8), near(512 // t8), near(768 // t8)} | ({1, 2, 3} if t8 > 128 else set())):
            ts = []
            for form in (65536, 0):
                L.mi355det_debug_set(7, spc + form)
                ts.append(min(timeit(lambda: ops.conv_wgrad(shape, x, dy, dw, workspace=ws)) for _ in range(3)))
            msg += f"  sp {spc:3d}: {ts[0]:6.1f} / {ts[1]:6.1f} us ({fl / ts[0] / 1e6:4.0f} TF)"
        L.mi355det_debug_set(7, 0)
        print(msg, flush=True)
