#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/ir_run.txt; : > $out
for nb in 0 2 3 4 6 12; do
  for cfg in cfg2 cfg3; do
  echo "== $cfg ir nb=$nb" >> $out
  AL_EXTRA_FLAGS=$((nb<<24)) python bench.py --config $cfg --cpu-events 0 --steps 10 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items() if v>0.03})
    elif 'rror' in l: print(l.strip())
" >> $out
  done
done
AL_EXTRA_FLAGS=$((3<<24)) python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> $out
