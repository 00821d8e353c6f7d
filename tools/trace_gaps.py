"""Device idle time and the largest kernels of one training step from a rocprofv3 kernel trace: the step = the launches between two
consecutive launches of a marker kernel (default `sgd_kernel`, the fused optimizer).  Idle = time with no kernel on any stream.
    python tools/trace_gaps.py <kernel_trace.csv> [marker] [min_gap_us] [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_kernel"
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = marks[-3], marks[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in step)
gaps, cur = [], iv[0][1]
for s, e, nm in iv[1:]:
    if s > cur:
        gaps.append((cur - t0, s - cur, nm))
    cur = max(cur, e)
idle = sum(g for _, g, _ in gaps)
print(f"step: {len(step)} launches, wall {(cur - t0) / 1e6:.2f} ms, device idle {idle / 1e3:.0f} us in {len(gaps)} gaps; gaps >= {min_gap:.0f} us:")
for at, g, nm in gaps:
    if g >= min_gap * 1e3:
        print(f"  at {at / 1e6:7.3f} ms  {g / 1e3:7.1f} us  before {re.sub(r'[(<].*', '', nm.replace('(anonymous namespace)::', '').replace('void ', ''))[:60]}")
agg = {}
for s, e, nm in iv:
    k = re.sub(r"\(anonymous namespace\)::", "", nm)
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*", "", k)[:80]
    x = agg.setdefault(k, [0, 0.0])
    x[0] += 1
    x[1] += (e - s) / 1e3
print(f"\nsummed kernel time {sum(v[1] for v in agg.values()) / 1e3:.2f} ms\n\n| kernel | launches | us |\n|---|---|---|")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"| `{k}` | {c} | {t:.1f} |")
