"""phase-stamp diagnostic of the shared-pixel-tile kernel: python tools/prof_dx.py cin cout hw [cfg=98]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd._lib import lib, ptr
dev = torch.device('cuda:0')
cin, cout, hw = [int(v) for v in sys.argv[1:4]]
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 98
n = 32
shape = ops.conv_shape(n, hw, hw, cin, cout, 3, 1)
x = torch.randn(n, hw, hw, cin, device=dev).bfloat16()
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
wf, wd = ops.pack_weights(shape, wt)
y = torch.empty(n, hw, hw, cout, device=dev, dtype=torch.bfloat16)
stats = torch.zeros(ops.conv_stats_rows(shape) + 64, 2, ops.cout_pad_of(cout), device=dev)
M = n * hw * hw
bm, bn, nw = (128, 128, 4) if cfg == 98 else (256, 128, 8)
blocks = ((M + bm - 1) // bm) * (cout // bn)
dbg = torch.zeros(blocks * nw * 8, device=dev, dtype=torch.int64)
lib().mi355det_debug_ptr(0, ptr(dbg))
for c in ((15, cfg) if cfg == 98 else (26, cfg)):
    lib().mi355det_debug_set(0, c)
    for _ in range(3):
        ops.conv_fwd(shape, x, wf, y, stats=stats)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_fwd(shape, x, wf, y, stats=stats)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"cfg {c}: {us:.1f} us  {2.0 * M * cout * 9 * cin / us / 1e6:.0f} TF/s")
lib().mi355det_debug_set(0, 0)
d = dbg.view(blocks, nw, 8).double()
steps = d[0, 0, 6].item()
print('k-steps per tile', steps, 'blocks', blocks)
for i, nm in enumerate(['vmcnt wait', 'barrier', 'DMA issue', 'fragment reads (drained)', 'MFMAs', 'loop total']):
    v = d[:, :, i]
    print(f'{nm:26s} mean/wave {v.mean().item():10.0f} cyc   per step {v.mean().item() / steps:7.0f}   (min {v.min().item():.0f} max {v.max().item():.0f})')
