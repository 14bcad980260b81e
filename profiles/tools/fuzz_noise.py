"""One-off fuzz of the arbitrary-length inverse real FFT behind powerlaw_psd_gaussian against the oracle (GPU)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiblelight_amd import ambience
from oracle import synth_oracle as orc

rng = np.random.default_rng(7)
lengths = [1, 2, 3, 4, 5, 7, 8, 9, 11, 13, 16, 17, 31, 97, 127, 128, 257, 509, 1000, 1001, 1021, 4093, 4096, 6000, 10007, 44100, 48000,
           65537, 99991, 131072, 240000] + [int(x) for x in rng.integers(2, 200000, 40)]
worst, bad = 0.0, 0
for n in lengths:
    for beta in (0, 1, 2, -1):
        rows = int(rng.integers(1, 4))
        try:
            got = ambience.powerlaw_psd_gaussian(beta, (rows, n), seed=int(n % 1000))
            want = orc.powerlaw_noise(beta, (rows, n), seed=int(n % 1000))
            den = np.sqrt(np.mean(want ** 2))
            err = np.sqrt(np.mean((got - want) ** 2)) / den if den > 0 else float(np.abs(got - want).max())
            worst = max(worst, err)
            if not err < 1e-4:
                bad += 1
                print("FAIL n", n, "beta", beta, "err", err, flush=True)
        except Exception as ex:
            bad += 1
            print("EXC n", n, "beta", beta, type(ex).__name__, str(ex)[:150], flush=True)
print("done: failures", bad, "worst rel rms", worst)
