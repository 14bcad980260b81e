# cfg5 at B = 16384 per library variant under profiles/tools/mfvariants/ (builds of csrc/al_quad16.h with other -D switches): per-stage ms
R=${GRAFT_REPO_ROOT:-.}; cd $R
for rep in 1 2; do
  for lib in profiles/tools/mfvariants/lib_*.so; do
    AUDIBLELIGHT_HIP_LIB=$PWD/$lib python3 bench.py --config ${CFG:-cfg5} --log2-block 14 --steps 10 --warmup 4 --repeats 3 --cpu-events 0 --cpu-workers 0 --dropin 0 --end-to-end 0 --other-configs 0 2>&1 | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('${CFG:-cfg5}', '$lib'.split('/')[-1], 'ms_per_step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['kernel_ms'].items() if v > 0.05})
    elif 'rror' in l: print('$lib', l.strip()[:200])"
  done
done
