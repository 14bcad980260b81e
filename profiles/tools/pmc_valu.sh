# Instruction counts (SQ_INSTS_*) per launch of the pipeline kernels on cfg2 and cfg5, one rocprofv3 --pmc run per config:
# the VALU issue floor of a kernel = SQ_INSTS_VALU x 4 cycles (one wave64 instruction per 4 cycles per SIMD) / (1024 SIMDs x 2.4 GHz)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for CFG in ${CFGS:-cfg2 cfg5}; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmcvalu -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --repeats 1 --cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 > $R/gpurun_out/pmcvalu.log 2>&1 || tail -3 $R/gpurun_out/pmcvalu.log
  f=$(find $R/gpurun_out/pmcvalu -name '*counter_collection.csv' | head -1)
  python3 - "$f" $CFG <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-44:]
    acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if any(s in k for s in ("forward_spectra", "spectral_mac", "block_synthesis", "mixdown")):
        extra = "   VALU floor %.3f ms" % (sum(v) / len(v) * 4 / (1024 * 2.4e9) * 1e3) if c == "SQ_INSTS_VALU" else ""
        print(f"{sys.argv[2]} {k:46s} {c:18s} {sum(v)/len(v):.5g}{extra}")
PY
  rm -rf $R/gpurun_out/pmcvalu
done
