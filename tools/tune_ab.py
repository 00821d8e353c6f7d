"""Derive an A/B tune record: every implicit-GEMM choice in `--from` (comma list of tile configurations) replaced by `--to`.
    python tools/tune_ab.py in.json out.json --from 44,45 --to 40
    python tools/tune_ab.py in.json out.json --wgrad8 512        # 3x3 weight gradients with Cout >= 512 onto the 256 x 256 phase-staggered kernel
Used for the same-box A/B of the 224 / 208-pixel tiles against the 256-pixel tile, and of the two weight-gradient kernels, inside the step
(profiles/r04_ab_results.md)."""
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
ap.add_argument("--wgrad8", type=int, default=0, help="3x3 weight gradients with at least this many output channels (multiples of 256) take the 256 x 256 "
                "kernel with one round of workgroups (0 = leave the weight gradients alone; then --from / --to apply)")
ap.add_argument("--rounds", type=int, default=256, help="--wgrad8: workgroups per launch to aim for (256 = one per CU)")
a = ap.parse_args()
frm = {int(v) for v in a.frm.split(",")}
ents = tune.to_entries(tune.loads(open(a.src).read()))
n = 0
out = []
for t, k, v in ents:
    if a.wgrad8:
        if t == "wgrad":
            m, cout, cin, ks, stride, _ = tune.wgrad_key_fields(k)
            if ks == 3 and cout >= max(256, a.wgrad8):
                t8 = ((cout + 255) // 256) * ((9 * cin + 255) // 256)
                sp = max(1, a.rounds // t8)
                while sp > 1 and not tune.wgrad_split_valid(m, sp):
                    sp -= 1
                print(f"  {cin} -> {cout} 3x3 s{stride}, {m} pixels: {v} -> form8 with {sp} splits")
                v, n = sp | tune.WGRAD_FORM8, n + 1
    elif t == "igemm" and v in frm:
        v, n = a.to, n + 1
    out.append((t, k, v))
with open(a.dst, "w") as f:
    f.write(tune.dumps(tune.from_entries(out)))
print(f"{n} of {len(ents)} entries changed")
