#!/bin/bash
# bench.py once per library variant built by build_defs.sh / build_variants.sh; prints per-stage ms.  $1 = extra bench args
cd "$(dirname "$0")/../.."
out=gpurun_out/variants.txt; : > $out
for rep in 1 2; do
for lib in profiles/tools/variants/lib_*.so; do
  echo "== $lib (pass $rep)" >> $out
  AUDIBLELIGHT_HIP_LIB=$PWD/$lib python bench.py --cpu-events 0 --steps 100 --end-to-end 0 --dropin 0 $1 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items() if v>0.03})
    elif 'rror' in l: print(l.strip())
" >> $out
done
done
cat $out
