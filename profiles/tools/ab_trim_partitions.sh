cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -2
for i in 1 2 3; do for T in 0 1; do
  AL_TRIM_PARTITIONS=$T python3 bench.py --config cfg3 --steps 30 --warmup 3 --repeats 3 --cpu-events 0 --cpu-workers 0 --dropin 0 --end-to-end 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trim=$T', round(d['ms_per_step'],3), d['timing']['ms_per_step_each_repeat'], 'fwd', round(d['roofline']['kernel_ms']['al_forward_spectra'],3), 'mac', round(d['roofline']['kernel_ms']['al_spectral_mac'],3))"
done; done
