"""Derive an A/B tune record: every implicit-GEMM choice in `--from` (comma list of tile configurations) replaced by `--to`.
    python tools/tune_ab.py in.json out.json --from 44,45 --to 40
Used for the same-box A/B of the 224 / 208-pixel tiles against the 256-pixel tile inside the step (profiles/r04_ab_results.md)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import tune  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("dst")
ap.add_argument("--from", dest="frm", default="44,45")
ap.add_argument("--to", type=int, default=40)
a = ap.parse_args()
frm = {int(v) for v in a.frm.split(",")}
ents = tune.to_entries(tune.loads(open(a.src).read()))
n = 0
out = []
for t, k, v in ents:
    if t == "igemm" and v in frm:
        v, n = a.to, n + 1
    out.append((t, k, v))
with open(a.dst, "w") as f:
    f.write(tune.dumps(tune.from_entries(out)))
print(f"{n} of {len(ents)} entries changed")
