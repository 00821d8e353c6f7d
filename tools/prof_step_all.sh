# Evidence for the bench.py step: kernel trace -> step / layer tables + kernel stats; three separate --pmc passes -> HBM traffic summary.
#   usage (on the GPU box, from the repo root): bash tools/prof_step_all.sh <tag>      -> gpurun_out/r4/<tag>_* (OUTDIR overrides r4)
set -e
TAG=${1:-r03}
O=gpurun_out/${OUTDIR:-r4}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-events"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -- $CMD > $O/prof_step.log 2>&1
python3 tools/step_table.py $O/prof_step 5 > $O/${TAG}_bench_step_table.md
python3 tools/layer_table.py $O/prof_step > $O/${TAG}_bench_layer_table.txt
python3 tools/summarize_rocprof.py $O/prof_step $O/${TAG}_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
grep "^{\"metric\"" $O/prof_step.log | tail -1 > $O/${TAG}_bench_line_under_rocprof.json
if [ "$2" != "nopmc" ]; then
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-events > $O/pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py 3 $O/${TAG}_pmc_summary.md $O/${TAG}_pmc_summary.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MfmaUtil
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MfmaUtil
fi
rm -rf $O/prof_step
