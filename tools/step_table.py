#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 --kernel-trace CSV of bench.py: only the last K timed steps (plan-build / autotune
launches are excluded), grouped by kernel family.

    python tools/step_table.py gpurun_out/prof5 5 > profiles/r01b_bench_step_table.md
"""
import collections
import csv
import glob
import re
import sys

path, K = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pack_batched_kernel" in r["Kernel_Name"]]
first = starts[-K]
sel = rows[first:]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(igemm\w*)<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(2).split(",")]
        epi = {"0": "stats(fwd+BN partials)", "1": "f32 head", "2": "residual(dgrad+skip)", "3": "plain(dgrad)", "4": "affine", "5": "dgrad+BN partial sums"}
        if m.group(1) == "igemm8_kernel":
            return f"igemm8 256x256x64 phase-staggered{' stream-K' if a[1] == 'true' else ''} epi={epi.get(a[0], a[0])}"
        if m.group(1) == "igemm_kernel":
            return f"igemm {int(a[0]) * int(a[2]) * 16}x{int(a[1]) * int(a[3]) * 16}x{a[4]} ring{a[5]} epi={epi.get(a[6], a[6])}"
        if m.group(1) == "igemm_dx_kernel":
            return f"igemm_dx {int(a[0]) * int(a[2]) * 16}x{int(a[1]) * int(a[3]) * 16}x64 (shared pixel tiles) epi={epi.get(a[4], a[4])}"
        return f"igemm_il {int(a[0]) * int(a[2]) * 16}x{int(a[1]) * int(a[3]) * 16}x{a[4]} epi={epi.get(a[5], a[5])}"
    return name.split("(")[0][:70]


agg = collections.OrderedDict()
for r in sel:
    f = family(r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(f, [0, 0.0, 1e30, 0.0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
busy = sum(a[1] for a in agg.values())
print(f"# per-step kernel table (last {K} steps of the run; plan build / autotune excluded)\n")
print(f"wall per step (first launch -> last completion): {(t1 - t0) / 1e6 / K:.3f} ms; summed kernel time per step: {busy / 1e3 / K:.3f} ms "
      f"(> wall where the two backward streams overlap)\n")
fw = [(n, t) for f, (n, t, _a, _b) in agg.items() if (f.startswith("igemm") and ("epi=stats" in f or "epi=f32" in f)) or f.startswith(("stem_l1_kernel", "stem_kernel<0>", "stem_kernel<1>"))]
nf, tf = sum(n for n, _ in fw), sum(t for _, t in fw)
print(f"forward convolution launches (igemm epi=stats + epi=f32 head + the stem forward kernels, the launches priced by bench.py's `roofline`): {nf / K:.0f} per step, "
      f"average duration {tf / nf:.2f} us, {tf / 1e3 / K:.3f} ms per step\n")
print("| kernel family | launches/step | ms/step | avg us | min us | max us | % of kernel time |\n|---|---|---|---|---|---|---|")
for f, (n, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{f}` | {n / K:.1f} | {t / 1e3 / K:.3f} | {t / n:.1f} | {mn:.1f} | {mx:.1f} | {100 * t / busy:.1f} |")
