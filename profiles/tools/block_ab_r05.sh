# Planner rule check (round 5): short IRs -- cfg4's 1 s RIR is 6 partitions of 8192 -- at B = 4096 (12 partitions, the 4096-point
# transform shape, clips of 47 blocks through the LDS-ring capsule loop) against the planner's B = 8192; cfg2 beside it.
R=${GRAFT_REPO_ROOT:-.}; cd $R
Q="--cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 --parity-events 0 --repeats 3"
show() { python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['kernel_ms'].items() if v > 0.02})
    elif 'rror' in l: print('$1', l.strip()[:200])"; }
for rep in 1 2; do
  for LB in 13 12; do python3 bench.py --config cfg4 --steps 100 --warmup 5 --log2-block $LB $Q 2>&1 | show "cfg4 log2_block=$LB"; done
  for LB in 13 12; do python3 bench.py --config cfg2 --steps 50 --warmup 5 --log2-block $LB $Q 2>&1 | show "cfg2 log2_block=$LB"; done
done
