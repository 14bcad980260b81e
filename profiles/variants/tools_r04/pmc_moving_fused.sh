# SQ counters of k_moving_fused (AL_FUSED_MOVING=1) on cfg3, one counter set per run (rocprofv3 --pmc, kernel trace only)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export AL_FUSED_MOVING=1
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVES" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum FETCH_SIZE"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcmf -- python3 $R/bench.py --config cfg3 --steps 2 --warmup 1 --repeats 1 --cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 > $R/gpurun_out/pmcmf.log 2>&1 || tail -3 $R/gpurun_out/pmcmf.log
  f=$(find $R/gpurun_out/pmcmf -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-40:]
    acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "moving_fused" in k or "forward_spectra" in k:
        print(f"{k:42s} {c:24s} {sum(v)/len(v):.5g}")
PY
  rm -rf $R/gpurun_out/pmcmf
done
