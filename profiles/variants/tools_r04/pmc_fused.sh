cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcf -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmcf -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-28:]
    acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "mac_synth" in k or "ir_spectra" in k or "mixdown" in k:
        print(f"{k:30s} {c:22s} {sum(v)/len(v):.4g}")
PY
  rm -rf $R/gpurun_out/pmcf
done
