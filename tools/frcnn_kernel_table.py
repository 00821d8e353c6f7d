"""Per-kernel device time of one Faster R-CNN training step from a rocprofv3 kernel trace of tools/bench_frcnn.py (training kernels only:
the window between the first and the last FlatSGD launch).   python tools/frcnn_kernel_table.py <kernel_trace.csv> [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "sgd" in r["Kernel_Name"].lower()]
lo, hi = opt[0], opt[-1]
rows = rows[lo:hi + 1]
steps = max(1, len([i for i in opt if True]) // max(1, len(set(r["Kernel_Name"] for r in rows if "sgd" in r["Kernel_Name"].lower()))) - 1)
agg = {}
for r in rows:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    nm = re.sub(r"^void ", "", nm)[:90]
    a = agg.setdefault(nm, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
busy = sum(a[1] for a in agg.values())
print(f"window: {steps} steps, {span / steps / 1e3:.2f} ms per step wall, {busy / steps / 1e3:.2f} ms per step summed kernel time\n")
print("| kernel | launches / step | us / step |\n|---|---|---|")
for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"| `{nm}` | {c / steps:.1f} | {t / steps:.1f} |")
