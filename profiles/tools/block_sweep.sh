Q="--cpu-events 0 --end-to-end 0 --dropin 0 --steps 20 --warmup 3"
for cfg in cfg5 cfg3 cfg4 cfg2; do for lb in 12 13 14; do
python3 bench.py --config $cfg --log2-block $lb $Q 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$cfg lb=$lb', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernel_ms'].items()})
"
done; done
