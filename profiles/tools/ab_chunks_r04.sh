# Round 4, on the final kernels: a cfg2 scene rendered in chunks of N events (accumulate -> synthesis of a chunk back to back, so that
# its output spectra could still be in the 256 MiB Infinity Cache when they are read) on 1..4 HIP streams, against one batch.
R=${GRAFT_REPO_ROOT:-.}; cd $R
Q="--steps 30 --warmup 5 --repeats 3 --cpu-events 0 --cpu-workers 0 --dropin 0 --end-to-end 0 --other-configs 0"
for gl in "0 1" "4 1" "4 2" "4 3" "8 1" "8 2" "8 3" "16 2" "2 4" "0 1"; do
  set -- $gl
  python3 bench.py --config ${CFG:-cfg2} --chunk-events $1 --lanes $2 $Q 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('${CFG:-cfg2} chunk_events=$1 lanes=$2', 'ms_per_step %.3f' % d['ms_per_step'])
    elif 'rror' in l: print(l.strip()[:200])"
done
