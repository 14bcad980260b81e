# FETCH_SIZE / WRITE_SIZE / L2 hit rate of the accumulate kernel under the current environment (AL_STATIC_MAC, AL_EXTRA_FLAGS)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcm -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 > $R/gpurun_out/pmcm.log 2>&1 || tail -5 $R/gpurun_out/pmcm.log
  f=$(find $R/gpurun_out/pmcm -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-44:]
    acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "mac" in k:
        print(f"{k:46s} {c:14s} {sum(v)/len(v):.4g}")
PY
  rm -rf $R/gpurun_out/pmcm
done
