#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/chunks_graph.txt; : > $out
for g in 0 1 2 3 4 8 16; do
  echo "== chunk_events=$g --graph" >> $out
  python bench.py --chunk-events $g --graph --cpu-events 0 --steps 10 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3))
    elif 'rror' in l: print(l.strip())
" >> $out
done
