"""ResNet stem, isolated: the direct kernel against im2col + GEMM (+ the max-pool both are followed by).   python tools/bench_rstem.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import ops
from object_detectors_amd._lib import check, lib, ptr, stream_ptr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
L = lib()
img = torch.rand((n, 3, 800, 800), device=dev)
wt = torch.randn((64, 3, 7, 7), device=dev) * 0.1
wp = torch.zeros((64, 160), dtype=torch.bfloat16, device=dev)
wp[:, :147] = wt.permute(0, 2, 3, 1).reshape(64, 147).bfloat16()
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
out = torch.empty((n, 400, 400, 64), dtype=torch.bfloat16, device=dev)
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
t = timeit(lambda: check(L.mi355det_resnet_stem_fwd(ptr(img), None, None, ptr(wp), ptr(sc), ptr(sh), 1, ptr(out), 64, n, 800, 800, stream_ptr()), "rstem"))
by = img.numel() * 4 + out.numel() * 2
print(f"resnet_stem_fwd (direct 7x7/2 + affine + ReLU) bs {n}: {t:7.1f} us   {by / 1e6:.0f} MB algorithmic -> {by / t / 1e6:.2f} TB/s; {2 * n * 160000 * 64 * 147 / t / 1e6:.0f} TFLOP/s")
for flag, what in ((3, "no fragment compute"), (5, "no halo fetch"), (9, "no halo LDS store"), (13, "no halo fetch + store")):
    tt = timeit(lambda: check(L.mi355det_resnet_stem_fwd(ptr(img), None, None, ptr(wp), ptr(sc), ptr(sh), flag, ptr(out), 64, n, 800, 800, stream_ptr()), "rstem"))
    print(f"   ablation ({what}): {tt:7.1f} us")
t2 = timeit(lambda: ops.im2col_nchw(img, 7, 2, 3, 160))
print(f"im2col_nchw alone (the matrix the GEMM route writes and re-reads: {n * 160000 * 160 * 2 / 1e6:.0f} MB): {t2:7.1f} us")
