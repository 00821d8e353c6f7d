"""single-layer launches for PMC collection: python tools/bench_one.py <tune> <cin> <cout> <k> <s> <hw> [iters]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd._lib import lib
dev = torch.device('cuda:0')
t, cin, cout, k, s, hw = [int(v) for v in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
n = 32
shape = ops.conv_shape(n, hw, hw, cin, cout, k, s)
x = torch.randn(n, hw, hw, cin, device=dev).bfloat16()
wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
wf, wd = ops.pack_weights(shape, wt)
y = torch.empty(n, shape.ho, shape.wo, cout, device=dev, dtype=torch.bfloat16)
dy = torch.randn(n, shape.ho, shape.wo, cout, device=dev).bfloat16()
dx = torch.empty_like(x)
dw = torch.zeros(cout, k * k * cin, device=dev)
stats = torch.zeros(ops.conv_stats_rows(shape) + 64, 2, ops.cout_pad_of(cout), device=dev)
lib().mi355det_debug_set(0, t)
for _ in range(iters):
    ops.conv_fwd(shape, x, wf, y, stats=stats)
    ops.conv_dgrad(shape, dy, wd, dx)
    ops.conv_wgrad(shape, x, dy, dw)
torch.cuda.synchronize()
