#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats CSV directory into a small committed summary under profiles/.

    python tools/summarize_rocprof.py gpurun_out/prof1 profiles/r01_bench_kernel_stats.md "command line"
"""
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    f = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\nsource: `{os.path.basename(f)}`; total kernel time {tot / 1e6:.2f} ms\n\n")
        o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            o.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {100 * float(r['TotalDurationNs']) / tot:.1f} |\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
