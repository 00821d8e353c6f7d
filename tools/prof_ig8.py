"""Phase stamps of the phase-staggered 256x256x64 conv kernel (diagnostic build igemm8_kernel<EPI_STATS, false, true>):

    python tools/prof_ig8.py > profiles/r02_igemm8_phase_stamps.txt

Per wave and k-step the loop has four phases, each (A) fragment reads + LDS-DMA issue, (B) counted vmcnt wait + barrier, (C) 16 MFMAs,
(D) trailing barrier.  Waves 0-3 (pixel half 0) and 4-7 (pixel half 1) share the four SIMDs pairwise (wave w and w+4) and run one barrier
apart, so one wave's (C) should coincide with its partner's (D)+(A)+(B).  The table prints the summed segments and the absolute stamps of one
k-step for the two waves of SIMD 0 of a few workgroups.

The diagnostic builds add stamp state to a loop that sits at the 256-VGPR cap: with the committed loop (fragment reads between the MFMAs) they
spill inside the loop and run 2-3x slower than the release build; the committed profile was taken on the loop form with the reads in the load
segments (234 VGPRs), where the one-stamp build is within 5 % of the release build."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops  # noqa: E402
from object_detectors_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
for (n, h, w, cin, cout, k, s) in [(32, 40, 40, 256, 512, 3, 1), (32, 20, 20, 512, 1024, 3, 1), (32, 80, 80, 128, 256, 3, 1)]:
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    x = torch.randn(n, h, w, cin, device=dev).bfloat16()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wf, wd = ops.pack_weights(shape, wt)
    y = torch.empty(n, shape.ho, shape.wo, cout, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(ops.conv_stats_rows(shape) + 64, 2, ops.cout_pad_of(cout), device=dev)
    flops = 2.0 * n * shape.ho * shape.wo * cout * cin * k * k
    lib().mi355det_debug_set(0, 40)

    def timed():
        for _ in range(3):
            ops.conv_fwd(shape, x, wf, y, stats=stats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv_fwd(shape, x, wf, y, stats=stats)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100
    us = timed()
    dbg = torch.zeros(64 * 8 * 24, dtype=torch.int64, device=dev)
    # ---- mode 2: one stamp per k-step (start of 12 consecutive k-steps of every wave): near-release timing
    lib().mi355det_debug_set(3, 2)
    lib().mi355det_debug_ptr(1, dbg.data_ptr())
    us_k = timed()
    lib().mi355det_debug_ptr(1, None)
    dk = dbg.cpu().numpy().astype(np.int64).reshape(64, 8, 24)
    caps = (dk[:, :, 5:17] & 0xFFFFFFFF).astype(np.int64)
    ok = (caps > 0).all(-1)                                  # waves whose tile had at least 16 k-steps
    per = np.diff(caps[ok], axis=-1)
    rt = (dk[:, :, 23] - dk[:, :, 22])[ok].astype(np.float64)            # 10 ns ticks over 11 k-steps
    ghz = (caps[ok][:, 11] - caps[ok][:, 0]) / (rt * 10.0)
    print(f"== conv {cin}->{cout} {k}x{k} @{shape.ho} bs {n}: release build {us:.1f} us ({flops / us / 1e6:.0f} TFLOP/s)")
    print(f"   cycle counter against the 100 MHz real-time counter over the same 11 k-steps: {ghz.mean():.2f} GHz (min {ghz.min():.2f} max {ghz.max():.2f}) "
          f"-> the MFMA peak at this clock is {2.5 * ghz.mean() / 2.4:.2f} PFLOP/s")
    print(f"   one stamp per k-step ({us_k:.1f} us): k-step period {per.mean():.0f} cycles (min {per.min()} max {per.max()}, median {np.median(per):.0f}); "
          f"a SIMD issues 2 x 64 MFMAs of 16 cycles = 2048 cycles per k-step -> MFMA pipe busy {100 * 2048 / per.mean():.0f} % inside the loop")
    tt = dk[:, :, 17:21].astype(np.float64)[ok] / 100.0          # us: workgroup start, loop start, loop end, epilogue done (stores issued and acknowledged)
    print(f"   per workgroup (us, mean over waves): prologue {np.mean(tt[:, 1] - tt[:, 0]):.2f} | main loop {np.mean(tt[:, 2] - tt[:, 1]):.2f} | "
          f"epilogue incl. store acknowledgement {np.mean(tt[:, 3] - tt[:, 2]):.2f} (max {np.max(tt[:, 3] - tt[:, 2]):.2f})")
    off = (caps[:, 4:, 0] - caps[:, :4, 0])
    off = off[ok[:, 4:] & ok[:, :4]]
    print(f"   start of the same k-step, wave w+4 minus wave w (SIMD partners): mean {off.mean():.0f} cycles (min {off.min()} max {off.max()}) = "
          f"{off.mean() / per.mean():.2f} of a k-step (1/8 = one barrier interval)")
    # ---- ablation builds: the same launch without the loop's LDS-DMA / fragment reads / MFMAs (garbage results, timing only)
    abl = {}
    for mode, nm in ((3, "no LDS-DMA"), (4, "no fragment reads"), (5, "no MFMAs"), (6, "32x32x16 MFMAs (same FLOPs)"), (7, "no trailing barriers"),
                     (8, "no barriers in the loop")):
        lib().mi355det_debug_set(3, mode)
        lib().mi355det_debug_ptr(1, dbg.data_ptr())
        abl[nm] = timed()
        lib().mi355det_debug_ptr(1, None)
    print("   ablations (whole launch, us): release %.1f | " % us + " | ".join(f"{k_} {v:.1f}" for k_, v in abl.items()))
    # ---- mode 1: phase stamps
    dbg.zero_()
    lib().mi355det_debug_set(3, 1)
    lib().mi355det_debug_ptr(1, dbg.data_ptr())
    us_p = timed()
    lib().mi355det_debug_ptr(1, None)
    lib().mi355det_debug_set(0, 0)
    d = dbg.cpu().numpy().astype(np.int64).reshape(64, 8, 24)
    steps = int(d[0, 0, 4])
    print(f"   phase-stamped build {us_p:.1f} us (17 stamps per k-step, each draining lgkmcnt; its loop also reloads spilled stamp state at the top of "
          f"every k-step, which shows up as ONE long barrier wait per k-step in the partner half); {steps} k-steps per tile")
    tot = d[:, :, :4].sum(-1).astype(np.float64)
    names = ["A reads + LDS-DMA issue", "B vmcnt wait + barrier", "C 16 MFMAs", "D trailing barrier"]
    for i, nm in enumerate(names):
        v = d[:, :, i].astype(np.float64) / steps / 4
        print(f"   {nm:26s} {v.mean():7.0f} cycles per phase  ({100 * d[:, :, i].sum() / tot.sum():4.1f} % of the loop; min {v.min():.0f} max {v.max():.0f})")
    print(f"   loop total per k-step: {tot.mean() / steps:.0f} cycles = 64 MFMAs; MFMA segment share {100 * d[:, :, 2].sum() / tot.sum():.1f} %")
    # overlap of the SIMD partners: fraction of wave 0's C segments (captured k-step) during which wave 4 is NOT in a C segment
    ov = []
    for wg in range(64):
        for w0 in range(4):
            a = d[wg, w0, 5:22] & 0xFFFFFFFF
            b = d[wg, w0 + 4, 5:22] & 0xFFFFFFFF
            if a[0] == 0 or b[0] == 0:
                continue
            ca = [(a[2 + 4 * q], a[3 + 4 * q]) for q in range(4)]
            cb = [(b[2 + 4 * q], b[3 + 4 * q]) for q in range(4)]
            la = sum(e - s_ for s_, e in ca)
            both = sum(max(0, min(e, e2) - max(s_, s2)) for s_, e in ca for s2, e2 in cb)
            ov.append(both / max(la, 1))
    print(f"   SIMD partners (wave w, w+4), captured k-step: {100 * np.mean(ov):.0f} % of a wave's MFMA-segment time coincides with the partner's MFMA "
          f"segment (0 % = perfectly complementary, n={len(ov)})")
    for wg in (0, 1):
        a0 = int(d[wg, 0, 5] & 0xFFFFFFFF)
        for wv in (0, 4):
            c = (d[wg, wv, 5:22] & 0xFFFFFFFF) - a0
            seg = "  ".join(f"P{q + 1}[A {c[1 + 4 * q] - c[4 * q]:4d} B {c[2 + 4 * q] - c[1 + 4 * q]:4d} C {c[3 + 4 * q] - c[2 + 4 * q]:4d} D {c[4 + 4 * q] - c[3 + 4 * q]:4d}]"
                            for q in range(4))
            print(f"   wg{wg} wave{wv}: k-step starts at +{c[0]:5d}; {seg}; C segments at " + " ".join(f"{c[2 + 4 * q]}-{c[3 + 4 * q]}" for q in range(4)))
