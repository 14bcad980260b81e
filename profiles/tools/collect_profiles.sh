cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc2_$c -- python $R/bench.py --steps 3 --warmup 1 --cpu-events 0 > $R/gpurun_out/pmc2_$c.log 2>&1; echo "$c rc=$?"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01b -- python $R/bench.py --steps 10 --warmup 2 --cpu-events 0 > $R/gpurun_out/prof_r01b.log 2>&1; echo "stats rc=$?"
