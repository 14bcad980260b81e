# HBM traffic of cfg3 through k_moving_fused (AL_FUSED_MOVING=1): the two PMC passes (separate runs), summarised like collect_profiles.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export AL_FUSED_MOVING=1
QUIET="--cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 --repeats 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r04q_pmc_$c -- python3 $R/bench.py --config cfg3 --steps 2 --warmup 1 $QUIET > $R/gpurun_out/r04q_pmc_$c.log 2>&1; echo "cfg3 fused $c rc=$?"
done
cd $R
F=$(find gpurun_out/r04q_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find gpurun_out/r04q_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 profiles/tools/summarise_pmc.py $F $W cfg3_fused/log2_block=13 gpurun_out/r04q_cfg3_fused_pmc_traffic.json
python3 profiles/tools/slim_pmc_csv.py $F gpurun_out/r04q_cfg3_fused_pmc_fetch_size.csv; python3 profiles/tools/slim_pmc_csv.py $W gpurun_out/r04q_cfg3_fused_pmc_write_size.csv
rm -rf gpurun_out/r04q_pmc_FETCH_SIZE gpurun_out/r04q_pmc_WRITE_SIZE
cat gpurun_out/r04q_cfg3_fused_pmc_traffic.json | head -40
