# Round 6: `python bench.py --gpus 2` -- the launcher itself (spawn_ranks), two rank processes -- on ONE MI355X (AL_BENCH_DEVICE=0
# puts both ranks on device 0; gloo for the collectives, RCCL refuses two ranks on one device) with the REAL kernels:
#   1. the headline mode with everything the N > 1 line carries since round 6: the PCIe-inclusive legs (batch driver + drop-in) run
#      by BOTH ranks at the same time between barriers, per-rank and aggregate rates, the gather held against one xGMI link;
#   2. a rank killed mid-run: how long the launcher needs to end the job, and with which exit code.
# Timings of (1) mean nothing as rates (one GPU and one PCIe link shared by two processes); what the record shows is that the legs
# run, agree on their barriers and report.  Usage: bash profiles/tools/two_ranks_one_gpu_r06.sh > gpurun_out/r06_two_ranks_one_gpu.txt
cd $GRAFT_REPO_ROOT
export AL_DIST_BACKEND=gloo AL_BENCH_DEVICE=0
python bench.py --gpus 2 --config cfg2 --steps 5 --warmup 2 --repeats 2 --cpu-events 0 --end-to-end 8 --dropin 4 > gpurun_out/two_r06_line.txt 2> gpurun_out/two_r06_err.txt
echo "exit code $?"
python - <<'PY'
import json
for l in open("gpurun_out/two_r06_line.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        strip = lambda x: {k: v for k, v in x.items() if k != "note"}
        print("n_gpus", d["n_gpus"], d["scaling"], "| ms/step by rank", d["timing"]["ms_per_step_by_rank_last_repeat"])
        print("end_to_end       ", strip(d["end_to_end"]))
        print("end_to_end_dropin", strip(d["end_to_end_dropin"]))
        print("gather           ", strip(d["gather"]))
        print("host_share       ", strip(d["host_share"]))
PY
tail -3 gpurun_out/two_r06_err.txt
echo "---- a rank killed mid-run"
T0=$(date +%s)
python bench.py --gpus 2 --config cfg2 --steps 2000000 --warmup 2 --repeats 1 --cpu-events 0 --end-to-end 0 --dropin 0 > gpurun_out/two_r06_dead.txt 2>&1 &
LAUNCHER=$!
sleep 45
VICTIM=""
for p in $(pgrep -P $LAUNCHER); do
  if tr '\0' '\n' < /proc/$p/environ | grep -qx "RANK=1"; then VICTIM=$p; fi
done
echo "launcher $LAUNCHER, rank 1 is pid $VICTIM, killing it $(( $(date +%s) - T0 )) s after the start"
TK=$(date +%s)
kill -9 $VICTIM
wait $LAUNCHER
CODE=$?
echo "launcher exit code $CODE, $(( $(date +%s) - TK )) s after the kill"
tail -2 gpurun_out/two_r06_dead.txt
