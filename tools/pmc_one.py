"""Print per-kernel PMC values of a rocprofv3 --pmc run of tools/bench_one.py: python tools/pmc_one.py <dir>"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0][:70]
    if "igemm" in n or "wgrad_kernel" in n:
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in agg.items():
    print(n, " ".join(f"{c}={sum(v[-3:]) / len(v[-3:]):.2f}" for c, v in sorted(cs.items())))
