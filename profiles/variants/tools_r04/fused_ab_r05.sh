# NOTE (round 6): reads AL_FUSED / AL_FUSED_MOVING, which left the library in commit 4b1d6c4: check out 4b1d6c4^ to reproduce
# profiles/r05c_fused_ab.txt; on a later tree both arms run the same kernels.
# Round 5, the time-boxed decision on the two opt-in fused kernels: same box, alternating, per-scene time.
#   k_mac_synthesis (csrc/al_fused.h, AL_FUSED=1)  on cfg4 (P = 6: the regime where its H / X re-reads per output block are smallest) and cfg2
#   k_moving_fused  (csrc/al_quad.h, AL_FUSED_MOVING=1) on cfg3
# Kill criterion: not >= 8 % faster per cfg4 scene -> both leave the product library.
R=${GRAFT_REPO_ROOT:-.}; cd $R
Q="--cpu-events 0 --cpu-workers 0 --end-to-end 0 --dropin 0 --other-configs 0 --parity-events 0 --repeats 3"
show() { python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms_per_step %.4f' % d['ms_per_step'], d['timing']['ms_per_step_each_repeat'], {k: round(v, 3) for k, v in d['roofline']['kernel_ms'].items() if v > 0.02}, d['config']['switches'])
    elif 'rror' in l: print('$1', l.strip()[:200])"; }
for rep in 1 2 3; do
  for CFG in cfg4 cfg2; do
    case $CFG in cfg4) S=100;; *) S=50;; esac
    for F in 0 1; do AL_FUSED=$F python3 bench.py --config $CFG --steps $S --warmup 5 $Q 2>&1 | show "$CFG AL_FUSED=$F"; done
  done
  for F in 0 1; do AL_FUSED_MOVING=$F python3 bench.py --config cfg3 --steps 20 --warmup 4 $Q 2>&1 | show "cfg3 AL_FUSED_MOVING=$F"; done
done
