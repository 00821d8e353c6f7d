set -e
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 300 python tools/layer_table_retina.py --body resnet50 --classes 91 --batch 16 > $O/retina_r50_layers.md 2>$O/retina_r50_layers.err
timeout -k 10 300 python tools/layer_table_retina.py --body resnet101 --classes 1204 --batch 8 > $O/retina_r101_layers.md 2>$O/retina_r101_layers.err
timeout -k 10 300 python tools/bench_retina.py > $O/retina_r50.json 2>/dev/null
timeout -k 10 300 python tools/bench_retina.py --body resnet101 --classes 1204 --batch 8 > $O/retina_r101.json 2>/dev/null
cat $O/retina_r50.json $O/retina_r101.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r50 -- python3 tools/bench_retina.py --steps 5 --warmup 2 > $O/prof_r50.log 2>&1
python3 tools/summarize_rocprof.py $O/prof_r50 $O/retina_r50_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/bench_retina.py --steps 5 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r101 -- python3 tools/bench_retina.py --body resnet101 --classes 1204 --batch 8 --steps 5 --warmup 2 > $O/prof_r101.log 2>&1
python3 tools/summarize_rocprof.py $O/prof_r101 $O/retina_r101_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/bench_retina.py --body resnet101 --classes 1204 --batch 8 --steps 5 --warmup 2"
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_r101_$c -- python3 tools/bench_retina.py --body resnet101 --classes 1204 --batch 8 --steps 3 --warmup 1 --no-fwd-only > $O/pmc_r101_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_r50_$c -- python3 tools/bench_retina.py --steps 3 --warmup 1 --no-fwd-only > $O/pmc_r50_$c.log 2>&1
done
python3 tools/pmc_summary.py 3 $O/retina_r101_pmc.md $O/retina_r101_pmc.json $O/pmc_r101_FETCH_SIZE $O/pmc_r101_WRITE_SIZE $O/pmc_r101_MfmaUtil
python3 tools/pmc_summary.py 3 $O/retina_r50_pmc.md $O/retina_r50_pmc.json $O/pmc_r50_FETCH_SIZE $O/pmc_r50_WRITE_SIZE $O/pmc_r50_MfmaUtil
rm -rf $O/prof_r50 $O/prof_r101 $O/pmc_r101_FETCH_SIZE $O/pmc_r101_WRITE_SIZE $O/pmc_r101_MfmaUtil $O/pmc_r50_FETCH_SIZE $O/pmc_r50_WRITE_SIZE $O/pmc_r50_MfmaUtil
cat $O/retina_r50_layers.md $O/retina_r101_layers.md
