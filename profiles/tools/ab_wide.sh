#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/wide.txt; : > $out
(cd profiles/tools && ./fft_probe2_m0 && ./fft_probe2_m1) >> $out 2>&1
for cfg in "cfg2 0" "cfg2 4" "cfg2 --log2-block=14 0" "cfg2 --log2-block=14 4" "cfg3 0" "cfg3 4" "cfg5 0" "cfg5 4"; do
  set -- $cfg; flags=${@: -1}; args="${@:1:$#-1}"
  echo "== $args AL_EXTRA_FLAGS=$flags" >> $out
  AL_EXTRA_FLAGS=$flags python bench.py --config $args --cpu-events 0 --steps 10 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items() if v>0.03})
    elif 'rror' in l: print(l.strip())
" >> $out
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> $out
