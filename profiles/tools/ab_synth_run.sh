for nb in 0 2 3 4 6 8 12 24; do
  echo "nb=$nb" >> gpurun_out/synth_run.txt
  AL_EXTRA_FLAGS=$((nb<<16)) python bench.py --cpu-events 0 --steps 20 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], {k:round(v,3) for k,v in d['roofline']['kernel_ms'].items()})
" >> gpurun_out/synth_run.txt
done
AL_EXTRA_FLAGS=$((4<<16)) python -m pytest tests -m gpu -x -q 2>&1 | tail -2 >> gpurun_out/synth_run.txt
python bench.py --cpu-workers -1 > gpurun_out/bench_allcores.json 2>&1
