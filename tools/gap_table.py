#!/usr/bin/env python3
"""Idle time between consecutive kernels of the forward pass (single stream) from a rocprofv3 --kernel-trace CSV of bench.py:
    python tools/gap_table.py gpurun_out/prof 5"""
import csv, glob, sys
path, K = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pack_batched_kernel" in r["Kernel_Name"]]
for si in starts[-K:]:
    seg = rows[si:]
    # forward = up to the first yolo loss kernel
    end = next(i for i, r in enumerate(seg) if "yolo_" in r["Kernel_Name"])
    f = seg[:end]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f)
    wall = int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])
    gaps = [int(f[i + 1]["Start_Timestamp"]) - int(f[i]["End_Timestamp"]) for i in range(len(f) - 1)]
    gaps_pos = [g for g in gaps if g > 0]
    print(f"forward: {len(f)} launches, wall {wall / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us "
          f"({100.0 * (wall - busy) / wall:.1f} %), mean gap {sum(gaps_pos) / max(1, len(gaps_pos)) / 1e3:.2f} us, max gap {max(gaps) / 1e3:.1f} us")
