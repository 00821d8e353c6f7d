"""Stream-K timeline of the phase-staggered conv kernel: per-workgroup s_memrealtime stamps (100 MHz).
tags: 1 start, 2 main loop of a segment done, 3 slab published, 4 partner slab visible, 5 slab added, 6 epilogue done"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd._lib import lib
dev = torch.device('cuda:0')
for (n, h, w, cin, cout, k, s) in [(32, 20, 20, 512, 1024, 3, 1), (32, 40, 40, 512, 256, 1, 1), (32, 80, 80, 128, 256, 3, 1)]:
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    x = torch.randn(n, h, w, cin, device=dev).bfloat16()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wf, wd = ops.pack_weights(shape, wt)
    y = torch.empty(n, shape.ho, shape.wo, cout, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(ops.conv_stats_rows(shape) + 64, 2, ops.cout_pad_of(cout), device=dev)
    dbg = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
    lib().mi355det_debug_set(0, 41)
    for _ in range(3): ops.conv_fwd(shape, x, wf, y, stats=stats)
    torch.cuda.synchronize()
    lib().mi355det_debug_ptr(1, dbg.data_ptr())
    ops.conv_fwd(shape, x, wf, y, stats=stats)
    torch.cuda.synchronize()
    lib().mi355det_debug_ptr(1, None)
    lib().mi355det_debug_set(0, 0)
    d = dbg.cpu().numpy().astype(np.uint64).reshape(256, 16)
    tags = (d >> np.uint64(56)).astype(int); t = (d & np.uint64((1 << 56) - 1)).astype(np.int64)
    t0 = t[:, 0][tags[:, 0] == 1].min()
    print(f"== {cin}->{cout} k{k} @{shape.ho}: kernel span {(t[tags > 0].max() - t0) / 100:.1f} us")
    for wg in list(range(0, 6)) + [100, 101, 254, 255]:
        print(f"  wg{wg:3d}: " + "  ".join(f"{tags[wg, i]}@{(t[wg, i] - t0) / 100:.1f}" for i in range(16) if tags[wg, i] > 0))
    dur = {k_: [] for k_ in ("slab_write", "wait", "slab_add", "epilogue", "drain", "epi_core", "epi_bar")}
    for wg in range(256):
        seq = [(tags[wg, i], t[wg, i]) for i in range(16) if tags[wg, i] > 0]
        for a, b in zip(seq, seq[1:]):
            if a[0] == 2 and b[0] == 7: dur["drain"].append(b[1] - a[1])
            if a[0] == 7 and b[0] == 3: dur["slab_write"].append(b[1] - a[1])
            if a[0] == 7 and b[0] == 4: dur["wait"].append(b[1] - a[1])
            if a[0] == 4 and b[0] == 5: dur["slab_add"].append(b[1] - a[1])
            if (a[0] == 7 or a[0] == 5) and b[0] == 8: dur["epi_core"].append(b[1] - a[1])
            if a[0] == 8 and b[0] == 6: dur["epi_bar"].append(b[1] - a[1])
    for k_, v in dur.items():
        if v: print(f"  {k_:10s}: n={len(v):3d} mean {np.mean(v) / 100:.2f} us  max {np.max(v) / 100:.2f} us")
