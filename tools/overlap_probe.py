"""Does an HBM-bound BatchNorm pass hide behind an MFMA-bound convolution when both run on separate streams?
Times N launches of each alone and both concurrently (two streams, optional priority).   python tools/overlap_probe.py"""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd._lib import check, lib, ptr, stream_ptr
dev = torch.device('cuda:0')
L = lib()
N = 20
def setup_conv(n, h, w, cin, cout, k, s, which):
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    x = torch.randn(n, h, w, cin, device=dev).bfloat16()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wf, wd = ops.pack_weights(shape, wt)
    y = torch.empty(n, shape.ho, shape.wo, cout, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(n, shape.ho, shape.wo, cout, device=dev).bfloat16()
    dx = torch.empty_like(x)
    dw = torch.zeros(cout, k * k * cin, device=dev)
    stats = torch.zeros(ops.conv_stats_rows(shape) + 64, 2, ops.cout_pad_of(cout), device=dev)
    L.mi355det_conv_autotune_mode(1)
    ops.conv_fwd(shape, x, wf, y, stats=stats); ops.conv_dgrad(shape, dy, wd, dx)
    L.mi355det_conv_autotune_mode(0)
    ws = torch.empty(L.mi355det_conv_wgrad_workspace(C.byref(shape)), device=dev, dtype=torch.uint8)
    L.mi355det_conv_wgrad_autotune(C.byref(shape), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), None)
    if which == 'wgrad': return lambda: ops.conv_wgrad(shape, x, dy, dw, workspace=ws)
    if which == 'dgrad': return lambda: ops.conv_dgrad(shape, dy, wd, dx)
    return lambda: ops.conv_fwd(shape, x, wf, y, stats=stats)
def setup_bn(c, hw, which):
    pixels = 32 * hw * hw
    z = torch.randn(pixels, c, device=dev).bfloat16(); g = torch.randn(pixels, c, device=dev).bfloat16(); out = torch.empty_like(z)
    ss = torch.cat([torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)]).to(dev)
    sums = torch.zeros(2 * c, device=dev); dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    if which == 'fwd': return lambda: check(L.mi355det_bn_act_fwd(ptr(z), c, ptr(ss), c, pixels, 0.1, None, 0, ptr(out), c, stream_ptr()))
    if which == 'reduce': return lambda: check(L.mi355det_bn_act_bwd_reduce(ptr(g), c, None, 0, ptr(z), c, ptr(ss), c, pixels, 0.1, ptr(sums), stream_ptr()))
    return lambda: check(L.mi355det_bn_act_bwd_apply(ptr(g), c, None, 0, ptr(z), c, ptr(ss), ptr(sums), None, c, pixels, 0.1, ptr(out), c, ptr(dg), ptr(db), stream_ptr()))
def run(fa, fb, sa, sb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    e0.record(st)
    if fa:
        sa.wait_event(e0)
        with torch.cuda.stream(sa):
            for _ in range(N): fa()
            ea = torch.cuda.Event(); ea.record(sa)
        st.wait_event(ea)
    if fb:
        sb.wait_event(e0)
        with torch.cuda.stream(sb):
            for _ in range(N): fb()
            eb = torch.cuda.Event(); eb.record(sb)
        st.wait_event(eb)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N
for prio in (0, -1):
    sa = torch.cuda.Stream(priority=0); sb = torch.cuda.Stream(priority=prio)
    for conv in [(32, 80, 80, 128, 256, 3, 1), (32, 40, 40, 256, 512, 3, 1)]:
        for cw in ('wgrad', 'dgrad'):
            for bn in ('apply', 'reduce'):
                fa = setup_conv(*conv, cw); fb = setup_bn(conv[4], conv[1], bn)
                for f in (fa, fb): f()
                ta, tb, tab = run(fa, None, sa, sb), run(None, fb, sa, sb), run(fa, fb, sa, sb)
                ta2, tb2, tab2 = run(fa, None, sa, sb), run(None, fb, sa, sb), run(fa, fb, sa, sb)
                print(f"bn-stream prio {prio}: {cw} {conv[3]}->{conv[4]}@{conv[1]} {min(ta, ta2):6.1f} us | bn_{bn} {min(tb, tb2):6.1f} us | both {min(tab, tab2):6.1f} us  (sum {min(ta, ta2) + min(tb, tb2):6.1f}, max {max(min(ta, ta2), min(tb, tb2)):6.1f})", flush=True)
