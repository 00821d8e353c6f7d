"""Derive a tune record without some weight-gradient entries: the plan build then times those shapes on the box (a locked record only pins the
shapes it holds).  Used to let the tuner re-decide shapes for which a new kernel form became available.
    python tools/tune_drop.py in.json out.json 20000 5000 1352      # the entries with these pixel counts (n * ho * wo)
    python tools/tune_drop.py in.json out.json all                   # every weight-gradient entry"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import tune  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
every = sys.argv[3:] == ["all"]
pix = set() if every else {int(v) for v in sys.argv[3:]}
ents = tune.to_entries(tune.loads(open(src).read()))
keep = [(t, k, v) for t, k, v in ents if not (t == "wgrad" and (every or tune.wgrad_key_fields(k)[0] in pix))]
with open(dst, "w") as f:
    f.write(tune.dumps(tune.from_entries(keep)))
print(f"{len(ents) - len(keep)} of {len(ents)} entries dropped")
