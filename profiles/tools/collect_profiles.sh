# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   per config (cfg2 cfg3 cfg4 cfg5): kernel-trace stats and the two PMC passes (separate runs, as MI355X_MICROARCH.md
#   prescribes), summarised into profiles/pmc_traffic.json under one key per config; then the bench lines, which read it.
#   gpurun -- 'bash profiles/tools/collect_profiles.sh r03a'          (optionally: ... r03a "cfg2 cfg4")
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; CONFIGS=${2:-"cfg2 cfg3 cfg4 cfg5"}
QUIET="--cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 --repeats 1"
rm -f $R/gpurun_out/${TAG}_pmc_traffic.json
for CFG in $CONFIGS; do
  case $CFG in cfg2|cfg4) PS=3; SS=50;; *) PS=2; SS=10;; esac
  case $CFG in cfg5) LB=14;; *) LB=13;; esac     # the planner's block size for the config (bench.py looks the traffic up under it)
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_${CFG}_pmc_$c -- python3 $R/bench.py --config $CFG --steps $PS --warmup 1 $QUIET > $R/gpurun_out/${TAG}_${CFG}_pmc_$c.log 2>&1; echo "$CFG $c rc=$?"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${CFG}_stats -- python3 $R/bench.py --config $CFG --steps $SS --warmup 3 $QUIET > $R/gpurun_out/${TAG}_${CFG}_stats.log 2>&1; echo "$CFG stats rc=$?"
  cd $R
  F=$(find gpurun_out/${TAG}_${CFG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find gpurun_out/${TAG}_${CFG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  V=$(find gpurun_out/${TAG}_${CFG}_pmc_SQ_INSTS_VALU -name '*counter_collection.csv' | head -1)
  python3 profiles/tools/summarise_pmc.py $F $W $CFG/log2_block=$LB gpurun_out/${TAG}_pmc_traffic.json $V > /dev/null; echo "$CFG summarise rc=$?"
  python3 profiles/tools/slim_pmc_csv.py $F gpurun_out/${TAG}_${CFG}_pmc_fetch_size.csv; python3 profiles/tools/slim_pmc_csv.py $W gpurun_out/${TAG}_${CFG}_pmc_write_size.csv
  cp $(find gpurun_out/${TAG}_${CFG}_stats -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_${CFG}_kernel_stats.csv
  python3 profiles/tools/slim_pmc_csv.py $V gpurun_out/${TAG}_${CFG}_pmc_valu_insts.csv
  rm -rf gpurun_out/${TAG}_${CFG}_pmc_FETCH_SIZE gpurun_out/${TAG}_${CFG}_pmc_WRITE_SIZE gpurun_out/${TAG}_${CFG}_pmc_SQ_INSTS_VALU gpurun_out/${TAG}_${CFG}_stats
  cd /tmp
done
cd $R
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json   # so the bench lines below carry roofline.traffic
python3 bench.py > gpurun_out/${TAG}_bench_cfg2.json 2> gpurun_out/${TAG}_bench_cfg2.err; echo "bench cfg2 rc=$?"
python3 bench.py --steps 20 --warmup 5 --cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 > gpurun_out/${TAG}_bench_cfg2_steps20.json 2>/dev/null; echo "bench cfg2 (driver's flags) rc=$?"
python3 bench.py --config cfg3 --steps 30 --warmup 3 --cpu-events 1 --cpu-workers 0 > gpurun_out/${TAG}_bench_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
python3 bench.py --config cfg5 --steps 30 --warmup 3 --cpu-events 1 --cpu-workers 0 --end-to-end 8 --dropin 0 > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
python3 bench.py --config cfg4 --steps 100 --warmup 3 --cpu-workers 0 > gpurun_out/${TAG}_bench_cfg4.json 2>/dev/null; echo "cfg4 rc=$?"
ls -la gpurun_out | grep ${TAG}
