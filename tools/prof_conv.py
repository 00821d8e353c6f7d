"""phase-stamp diagnostic of the igemm main loop: python tools/prof_conv.py cin cout k s hw"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd._lib import lib, ptr
dev = torch.device('cuda:0')
cin, cout, k, s, hw = [int(v) for v in sys.argv[1:6]]
n = 32
shape = ops.conv_shape(n, hw, hw, cin, cout, k, s)
x = torch.randn(n, hw, hw, cin, device=dev).bfloat16()
wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
wf, wd = ops.pack_weights(shape, wt)
y = torch.empty(n, shape.ho, shape.wo, cout, device=dev, dtype=torch.bfloat16)
stats = torch.zeros(ops.conv_stats_rows(shape) + 64, 2, ops.cout_pad_of(cout), device=dev)
M = n * shape.ho * shape.wo
blocks = (M // 128) * (cout // 128)
dbg = torch.zeros(blocks * 4 * 8, device=dev, dtype=torch.int64)
lib().mi355det_debug_ptr(0, ptr(dbg))
lib().mi355det_debug_set(0, 99)
for _ in range(3):
    ops.conv_fwd(shape, x, wf, y, stats=stats)
torch.cuda.synchronize()
d = dbg.view(blocks, 4, 8).double()
ks = k * k * cin // 64
print('ksteps', ks, 'blocks', blocks)
names = ['vmcnt wait', 'barrier', 'stage issue', 'ds_read+mfma', 'prologue', 'loop total']
for i, nm in enumerate(names):
    v = d[:, :, i]
    per = v.mean().item() / (ks if i < 4 else 1)
    print(f'{nm:14s} mean/wave {v.mean().item():9.0f} cyc  per-step {per:7.0f}   (min {v.min().item():.0f} max {v.max().item():.0f})')
raw = dbg.view(blocks, 4, 8)
for nm, idx, hi in (('start->p1 (toff, bid)', 6, True), ('p1->p2 (descriptors+rows)', 6, False), ('p2->p3 (b offsets, lgkm)', 7, True), ('p3->p4 (barrier)', 7, False)):
    v = (raw[:, :, idx] >> 32) if hi else (raw[:, :, idx] & 0xFFFFFFFF)
    print(f'{nm:28s} mean {v.double().mean().item():8.0f}')
