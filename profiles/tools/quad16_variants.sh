# bench.py per library variant under profiles/tools/mfvariants/ (builds of csrc/al_transforms.hip with other -D switches): per-stage ms
#   CFGS="cfg5 cfg2" (default) ; cfg5 runs at the planner's block size
R=${GRAFT_REPO_ROOT:-.}; cd $R
for CFG in ${CFGS:-cfg5 cfg2}; do
  case $CFG in cfg2|cfg4) S=50;; *) S=10;; esac
  for rep in 1 2; do
    for lib in profiles/tools/mfvariants/lib_*.so; do
      AUDIBLELIGHT_HIP_LIB=$PWD/$lib python3 bench.py --config $CFG $XARGS --steps $S --warmup 4 --repeats 3 --cpu-events 0 --cpu-workers 0 --dropin 0 --end-to-end 0 --other-configs 0 2>&1 | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$CFG', '$lib'.split('/')[-1], 'ms_per_step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['kernel_ms'].items() if v > 0.05})
    elif 'rror' in l: print('$lib', l.strip()[:200])"
    done
  done
done
