#!/bin/bash
# bench.py once per library variant built by build_variants.sh; prints per-stage ms.
cd "$(dirname "$0")/../.."
out=gpurun_out/variants.txt; : > $out
for lib in profiles/tools/variants/lib_*.so; do
  echo "== $lib" >> $out
  AUDIBLELIGHT_HIP_LIB=$PWD/$lib python bench.py --cpu-events 0 --steps 20 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items() if v>0.03})
    elif 'rror' in l: print(l.strip())
" >> $out
done
