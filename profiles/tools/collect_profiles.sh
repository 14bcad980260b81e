# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, the two PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) and the bench lines.
#   gpurun -- 'bash profiles/tools/collect_profiles.sh r02a'
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-r02}
QUIET="--cpu-events 0 --end-to-end 0 --dropin 0"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 $QUIET > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1; echo "$c rc=$?"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 50 --warmup 5 $QUIET > $R/gpurun_out/${TAG}_stats.log 2>&1; echo "stats rc=$?"
cd $R
F=$(find gpurun_out/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find gpurun_out/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 profiles/tools/summarise_pmc.py $F $W cfg2/log2_block=13 gpurun_out/${TAG}_pmc_traffic.json; echo "summarise rc=$?"
cp $F gpurun_out/${TAG}_pmc_fetch_size.csv; cp $W gpurun_out/${TAG}_pmc_write_size.csv
cp $(find gpurun_out/${TAG}_stats -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats.csv
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json   # so the bench lines below carry roofline.traffic
python3 bench.py > gpurun_out/${TAG}_bench_cfg2.json 2> gpurun_out/${TAG}_bench_cfg2.err; echo "bench cfg2 rc=$?"
python3 bench.py --config cfg3 --steps 30 --warmup 3 --cpu-events 1 > gpurun_out/${TAG}_bench_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
python3 bench.py --config cfg5 --steps 30 --warmup 3 --cpu-events 1 --end-to-end 0 > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
python3 bench.py --config cfg4 --steps 100 --warmup 3 --cpu-events 2 > gpurun_out/${TAG}_bench_cfg4.json 2>/dev/null; echo "cfg4 rc=$?"
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_stats
ls -la gpurun_out | grep ${TAG}
