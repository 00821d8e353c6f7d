"""RoIAlign of the Faster R-CNN box head in isolation: 2048 RoIs (512 per image) on the four bf16 NHWC pyramid levels of a batch of 4 at
800 px, 256 channels, 7x7, sampling_ratio 2; forward and backward, separable form (the per-sample form of round 3's A/B, 432 / 1326 us against 177 / 526, is no longer selectable).
    python tools/bench_roi_align.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, c = 4, 256
feats = [torch.randn((n, s, s, c), device=dev, generator=g).bfloat16() for s in (200, 100, 50, 25)]
scales = [0.25, 0.125, 0.0625, 0.03125]
K = 512 * n
ctr = torch.rand((K, 2), device=dev, generator=g) * 700 + 50
wh = torch.exp(torch.rand((K, 2), device=dev, generator=g) * 3.2 + 2.8)          # 16 .. 400 px sides, log-uniform
rois = torch.cat([(torch.arange(K, device=dev) // 512).float()[:, None], (ctr - wh / 2).clamp(0, 800), (ctr + wh / 2).clamp(0, 800)], 1)
go = torch.randn((K, c, 7, 7), device=dev, generator=g)
def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
f = timeit(lambda: ops.roi_align_nhwc(feats, rois, 7, scales, 2, False, 2, 5))
b = timeit(lambda: ops.roi_align_nhwc(feats, rois, 7, scales, 2, False, 2, 5, grad_out=go))
z = timeit(lambda: [torch.zeros((x.shape[0], x.shape[1], x.shape[2], c), device=dev) for x in feats])
print(f"separable form: forward {f:.1f} us, backward {b:.1f} us (of which zero-fill of the gradient maps {z:.1f} us)")
