# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, the two PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) and the bench lines.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-r01e}
python -m pytest $R/tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_$c -- python $R/bench.py --steps 3 --warmup 1 --cpu-events 0 > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1; echo "$c rc=$?"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python $R/bench.py --steps 20 --warmup 3 --cpu-events 0 > $R/gpurun_out/${TAG}_stats.log 2>&1; echo "stats rc=$?"
cd $R
python bench.py > gpurun_out/${TAG}_bench_cfg2.json 2> gpurun_out/${TAG}_bench_cfg2.err; echo "bench cfg2 rc=$?"
python bench.py --config cfg3 --steps 10 --warmup 3 --cpu-events 1 > gpurun_out/${TAG}_bench_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
python bench.py --config cfg5 --steps 10 --warmup 3 --cpu-events 1 > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
python bench.py --config cfg4 --steps 10 --warmup 2 --cpu-events 2 > gpurun_out/${TAG}_bench_cfg4.json 2>/dev/null; echo "cfg4 rc=$?"
