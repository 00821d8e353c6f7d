"""Top kernels per training step from a rocprofv3 --kernel-trace CSV of one of the tools/bench_*.py scripts (steps delimited by the fused
optimizer launch):  python tools/step_kernels.py <trace dir> <steps>"""
import collections, csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
a, b = idx[-steps - 1], idx[-1]
sel = rows[a + 1:b + 1]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6 / steps
print(f"per step: wall {wall:.2f} ms, summed kernel time {sum(dur(r) for r in sel) / steps:.2f} ms, launches {len(sel) / steps:.0f}")
agg, cnt = collections.Counter(), collections.Counter()
for r in sel:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    agg[k] += dur(r) / steps
    cnt[k] += 1
for k, v in agg.most_common(25):
    print(f"{v:7.3f} ms  x{cnt[k] / steps:5.1f}  {k}")
