# cfg3 with the IR-transform-fused moving accumulate (AL_FUSED_MOVING=1) against the default path over stored IR spectra
# (AL_FUSED_MOVING=0), same box, alternating:   gpurun -- 'bash profiles/tools/cfg3_fused_ab.sh r04b'
R=${GRAFT_REPO_ROOT:-.}; TAG=${1:-r04}; cd $R
Q="--config cfg3 --steps 20 --warmup 4 --repeats 3 --cpu-events 0 --cpu-workers 0 --dropin 0 --end-to-end 0 --other-configs 0"
for i in 1 2; do
  for F in 1 0; do
    AL_FUSED_MOVING=$F python3 bench.py $Q 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('AL_FUSED_MOVING=$F', 'ms_per_step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['kernel_ms'].items()})"
  done
done
