# Round 4: B = 16384 in four 4096-point transforms (csrc/al_quad16.h) against the default B = 8192, and against the one- / two-
# transform kernels at B = 16384; same box, alternating:  gpurun -- 'bash profiles/tools/quad16_ab.sh'
R=${GRAFT_REPO_ROOT:-.}; cd $R
Q="--steps 10 --warmup 4 --repeats 3 --cpu-events 0 --cpu-workers 0 --dropin 0 --end-to-end 0 --other-configs 0"
run() {  # label, config, env..., -- extra args
  label=$1; cfg=$2; shift 2
  env "$@" python3 bench.py --config $cfg $Q $EXTRA 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$label', 'ms_per_step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['kernel_ms'].items() if v > 0.05})"
}
for i in 1 2; do
  for cfg in ${CONFIGS:-cfg5 cfg2}; do
    EXTRA="" run "lb13" $cfg AL_X=0
    EXTRA="--log2-block 14" run "lb14_quad16" $cfg AL_QUAD16=1
    [ $i = 1 ] && EXTRA="--log2-block 14" run "lb14_one_transform" $cfg AL_QUAD16=0
  done
done
