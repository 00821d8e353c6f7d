"""Per-kernel totals of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [name substring ...]
Prints calls, total ms, average us for the kernels whose (demangled) name contains one of the substrings (all kernels if none)."""
import csv, glob, os, sys
d = sys.argv[1]
subs = sys.argv[2:]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows:
    n = r["Name"]
    if subs and not any(s in n for s in subs):
        continue
    print(f"{int(r['Calls']):6d} calls {float(r['TotalDurationNs']) / 1e6:9.3f} ms  avg {float(r['AverageNs']) / 1e3:8.1f} us  {n[:150]}")
