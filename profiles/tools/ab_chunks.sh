#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/chunks.txt; : > $out
for gl in "0 1" "1 1" "2 1" "2 2" "4 1" "4 2" "8 1" "8 2" "16 2" "4 4"; do
  set -- $gl
  echo "== chunk_events=$1 lanes=$2" >> $out
  python bench.py --chunk-events $1 --lanes $2 --cpu-events 0 --steps 10 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3))
    elif 'rror' in l: print(l.strip())
" >> $out
done
