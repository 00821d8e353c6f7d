#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: per queue (stream) the kernels in start order with start offset,
duration and the gap to the previous kernel of the same queue; kernels longer than --min-us only.  Used to tell a slow kernel from a kernel
that waited (profiles/r04_retinanet_r101_timeline.md: the "FPN dgrad 4.8 ms" line of round 3).
    python tools/trace_timeline.py <rocprof dir> [--step-marker pack_batched_kernel] [--min-us 150] [--step -2]"""
import argparse, csv, glob
ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--step-marker", default="pack_batched_kernel")
ap.add_argument("--min-us", type=float, default=150.0)
ap.add_argument("--step", type=int, default=-2)
ap.add_argument("--contains", default=None, help="pick the LAST complete step that launches a kernel with this substring (e.g. wgrad_kernel: a training step)")
a = ap.parse_args()
rows = list(csv.DictReader(open(glob.glob(a.path + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.step_marker in r["Kernel_Name"]]
# a step starts at a weight-pack launch that is at least 1 ms after the previous one (trainable + frozen packs come in pairs)
starts = [m for j, m in enumerate(marks) if j == 0 or int(rows[m]["Start_Timestamp"]) - int(rows[marks[j - 1]]["Start_Timestamp"]) > 1_000_000]
if a.contains:
    cand = [j for j in range(len(starts) - 1) if any(a.contains in r["Kernel_Name"] for r in rows[starts[j]:starts[j + 1]])]
    a.step = cand[-1] - len(starts)
s0, s1 = starts[a.step], starts[a.step + 1] if a.step + 1 < 0 else len(rows)
seg = rows[s0:s1]
t0 = int(seg[0]["Start_Timestamp"])
qkey = "Queue_Id" if "Queue_Id" in seg[0] else "Stream_Id"
queues = {}
for r in seg:
    queues.setdefault(r[qkey], []).append(r)
print(f"step: {len(seg)} launches, {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms, queues: " + ", ".join(f"{q}: {len(v)} launches, busy {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in v) / 1e6:.2f} ms" for q, v in queues.items()))
# device idle = wall - union of all kernel intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for st, en in iv[1:]:
    if st > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = st, en
    else:
        cur_e = max(cur_e, en)
busy += cur_e - cur_s
wall = int(seg[-1]["End_Timestamp"]) - t0
print(f"device busy (union of kernel intervals) {busy / 1e6:.2f} ms of {wall / 1e6:.2f} ms: idle {(wall - busy) / 1e6:.2f} ms")
by = {}
for r in seg:
    n = r["Kernel_Name"]
    for pre in ("void ", "(anonymous namespace)::"):
        if n.startswith(pre):
            n = n[len(pre):]
    n = n.split("(")[0][:70]
    d = by.setdefault(n, [0, 0])
    d[0] += 1
    d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("\n| kernel | launches | total us |\n|---|---|---|")
for n, (c, tns) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"| {n} | {c} | {tns / 1e3:.0f} |")
for q, v in queues.items():
    print(f"\n## queue {q}\n\n| start ms | dur us | gap before us | workgroups | kernel |\n|---|---|---|---|---|")
    prev_end = None
    for r in v:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if prev_end is None else (st - prev_end) / 1e3
        prev_end = max(prev_end or 0, en)
        if (en - st) / 1e3 >= a.min_us or gap >= a.min_us:
            wg = int(r.get("Grid_Size", 0) or 0) // max(1, int(r.get("Workgroup_Size", 1) or 1))
            print(f"| {(st - t0) / 1e6:.3f} | {(en - st) / 1e3:.0f} | {gap:.0f} | {wg} | {r['Kernel_Name'].replace('void (anonymous namespace)::', '')[:90]} |")
