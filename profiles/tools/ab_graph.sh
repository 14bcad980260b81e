cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/graph.txt
for cfg in cfg1 cfg4 cfg2; do for g in "" "--graph"; do
echo "== $cfg $g" >> gpurun_out/graph.txt
python bench.py --config $cfg $g --cpu-events 0 --steps 50 --warmup 5 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items()})
    elif 'rror' in l: print(l.strip())
" >> gpurun_out/graph.txt
done; done
