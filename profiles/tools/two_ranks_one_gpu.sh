# Two rank processes on ONE MI355X (both LOCAL_RANK=0; gloo for the collectives, RCCL refuses two ranks on one device): the
# real kernels under the N = 2 control flow of bench.py -- sharding, per-repeat MAX, end-of-job gather and its self-validation
# (rank 0 re-renders rank 1's scene with the real kernels and compares bit for bit).  Timings mean nothing (one GPU shared).
cd $GRAFT_REPO_ROOT
export AL_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0
for MODE in "--config cfg2" "--config cfg4 --total-scenes 6" "--config cfg5 --shard capsules --scale 0.25" "--config cfg5 --shard capsules --scale 0.5 --log2-block 14"; do
  TAG=$(echo $MODE | tr -d ' -')
  RANK=1 python bench.py --gpus 2 $MODE --steps 5 --warmup 2 --repeats 2 --cpu-events 0 > gpurun_out/two_rank1_$TAG.txt 2>&1 &
  RANK=0 python bench.py --gpus 2 $MODE --steps 5 --warmup 2 --repeats 2 --cpu-events 0 > gpurun_out/two_rank0_$TAG.txt 2>&1
  wait
  python - <<PY
import json
for l in open("gpurun_out/two_rank0_$TAG.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("$MODE", "| n_gpus", d["n_gpus"], d["scaling"], "| ms/step by rank", d["timing"]["ms_per_step_by_rank_last_repeat"], "| gather", {k: v for k, v in d["gather"].items() if k != "note"}, "| collectives", d.get("collectives_ms", {}).get("allreduce_event_levels"))
PY
done
