# cfg3 once per timing build of k_moving_fused (profiles/tools/mfvariants/lib_*.so from -DAL_MF_SKIP=...): per-stage ms
R=${GRAFT_REPO_ROOT:-.}; cd $R
Q="--config cfg3 --steps 10 --warmup 3 --repeats 1 --cpu-events 0 --cpu-workers 0 --dropin 0 --end-to-end 0 --other-configs 0"
for lib in profiles/tools/mfvariants/lib_*.so; do
  AUDIBLELIGHT_HIP_LIB=$PWD/$lib python3 bench.py $Q 2>&1 | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', 'ms_per_step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['kernel_ms'].items() if v > 0.05})
    elif 'rror' in l: print('$lib', l.strip()[:200])"
done
