# (round 4 experiment, see profiles/r04g_cfg5_mac_kt8_ab.txt; the kernel it switched on is not in the library any more)
# cfg5's accumulate: the capsule loop with k-tiles of 8 blocks against the tile kernel
# k_spectral_mac<12,12,2,KSPLIT> (AL_STATIC_MAC_MAX_P=21), same box, alternating:  gpurun -- 'bash profiles/tools/cfg5_mac_ab.sh'
R=${GRAFT_REPO_ROOT:-.}; cd $R
Q="--config cfg5 --steps 10 --warmup 4 --repeats 3 --cpu-events 0 --cpu-workers 0 --dropin 0 --end-to-end 0 --other-configs 0"
for i in 1 2; do
  for MAXP in 24 21; do
    AL_STATIC_MAC_MAX_P=$MAXP python3 bench.py $Q 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('AL_STATIC_MAC_MAX_P=$MAXP', 'ms_per_step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['roofline']['kernel_ms'].items() if v > 0.05})"
  done
done
