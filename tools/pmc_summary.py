#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py (one counter set per pass, as MI355X_MICROARCH.md prescribes) over the last K steps.

    python tools/pmc_summary.py K out.md out.json dir_fetch dir_write dir_mfma

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced
reads at 64 B, so it is doubled (guide, section HBM).  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM).
"""
import collections
import csv
import glob
import json
import re
import sys

K = int(sys.argv[1])
out_md, out_json = sys.argv[2], sys.argv[3]
dirs = sys.argv[4:]


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(igemm\w*)<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(2).split(",")]
        kind = m.group(1)
        e = a[0] if kind == "igemm8_kernel" else a[6] if kind == "igemm_kernel" else (a[4] if kind == "igemm_dx_kernel" else a[5])
        return "igemm " + {"0": "fwd (BN partial stats)", "1": "fwd (fp32 head)", "2": "dgrad (+skip)", "3": "dgrad", "4": "fwd (affine)"}.get(e, e)
    if name.startswith("stem_kernel<"):
        return name.split("(")[0][:60]        # the four modes of csrc/stem_kernels.hip are different kernels
    return name.split("(")[0].split("<")[0][:60]


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # dispatch ids of the step starts (one weight-pack launch per step)
    packs = sorted({int(r["Dispatch_Id"]) for r in rows if "pack_batched_kernel" in r["Kernel_Name"]})
    first = packs[-K]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in rows:
        if int(r["Dispatch_Id"]) < first:
            continue
        a = agg[family(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


tot = collections.defaultdict(dict)
for d in dirs:
    for fam, cs in load(d).items():
        for c, (n, v) in cs.items():
            tot[fam][c] = (n, v)
res = {}
lines = ["| kernel family | launches/step | HBM read MB/launch (FETCH_SIZE x2) | HBM write MB/launch | MfmaUtil % (launch mean) |", "|---|---|---|---|---|"]
for fam, cs in sorted(tot.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[1])):
    n = max(v[0] for v in cs.values())
    rd = 2.0 * cs["FETCH_SIZE"][1] * 1024 / cs["FETCH_SIZE"][0] / 1e6 if "FETCH_SIZE" in cs else None
    wr = cs["WRITE_SIZE"][1] * 1024 / cs["WRITE_SIZE"][0] / 1e6 if "WRITE_SIZE" in cs else None
    mu = cs["MfmaUtil"][1] / cs["MfmaUtil"][0] if "MfmaUtil" in cs else None
    res[fam] = {"launches_per_step": n / K, "read_MB_per_launch": rd, "write_MB_per_launch": wr, "mfma_util_pct": mu}
    f = lambda v, p=2: "-" if v is None else f"{v:.{p}f}"
    lines.append(f"| `{fam}` | {n / K:.1f} | {f(rd)} | {f(wr)} | {f(mu, 1)} |")
open(out_md, "w").write("# PMC summary (separate --pmc passes; last %d steps; plan build excluded)\n\n" % K + "\n".join(lines) + "\n")
json.dump(res, open(out_json, "w"), indent=1)
print("\n".join(lines[:12]))
