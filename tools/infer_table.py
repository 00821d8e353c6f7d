"""Kernel table of the eval-mode forward from a rocprofv3 kernel trace of tools/prof_infer.py (last 5 forwards)."""
import collections, csv, glob, re, sys
path = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "stem_kernel<1>" in r["Kernel_Name"]]
K = 5
sel = rows[starts[-K]:]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
agg = collections.OrderedDict()
for r in sel:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0][:80]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = sum(a[1] for a in agg.values())
gaps = (t1 - t0) / 1e3 - busy
print(f"# eval-mode YOLOv3-D53 forward, bs 32 / 640 px, frozen weights: last {K} forwards\n")
print(f"wall per forward (first launch -> last completion, incl. the gap to the next forward): {(t1 - t0) / 1e6 / K:.3f} ms; kernel time {busy / 1e3 / K:.3f} ms; "
      f"idle between kernels {gaps / 1e3 / K:.3f} ms; {sum(a[0] for a in agg.values()) / K:.0f} launches per forward\n")
print("| kernel | launches/forward | ms/forward | avg us |\n|---|---|---|---|")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{n}` | {c / K:.1f} | {t / 1e3 / K:.3f} | {t / c:.1f} |")
