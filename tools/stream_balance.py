"""Backward-pass stream balance from a rocprofv3 kernel trace of bench.py: busy time per stream, both-busy, idle, over the backward window of the
last step.    python tools/stream_balance.py <trace dir>"""
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_batched_kernel" in r["Kernel_Name"]]
last = rows[idx[-2]:idx[-1]]
b0 = next(i for i, r in enumerate(last) if "yolo_" in r["Kernel_Name"])
bw = last[b0:]
t0, t1 = int(bw[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in bw)
streams = collections.defaultdict(list)
for r in bw:
    streams[r["Stream_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
def union(iv):
    iv = sorted((a, b) for a, b, *_ in iv)
    out, cs, ce = [], *iv[0]
    for s, e in iv[1:]:
        if s > ce:
            out.append((cs, ce)); cs, ce = s, e
        else:
            ce = max(ce, e)
    out.append((cs, ce))
    return out
def length(u): return sum(b - a for a, b in u)
def inter(u, v):
    i = j = 0; tot = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b: tot += b - a
        if u[i][1] < v[j][1]: i += 1
        else: j += 1
    return tot
us = {k: union(v) for k, v in streams.items()}
print(f"backward window {(t1 - t0) / 1e6:.3f} ms")
for k, u in us.items():
    names = collections.Counter(n.split("(")[0][-40:] for _a, _b, n in streams[k]).most_common(3)
    print(f"stream {k}: busy {length(u) / 1e6:.3f} ms ({len(streams[k])} kernels; mostly {[n for n, _ in names]})")
ks = list(us)
if len(ks) >= 2:
    both = inter(us[ks[0]], us[ks[1]])
    allu = union([iv for k in ks for iv in us[k]])
    print(f"both busy {both / 1e6:.3f} ms; any busy {length(allu) / 1e6:.3f} ms; idle {(t1 - t0 - length(allu)) / 1e6:.3f} ms")
