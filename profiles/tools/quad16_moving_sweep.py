"""Round 4: MOVING events with long IRs at B = 8192 against B = 16384 (quad-tile transforms, csrc/al_quad16.h): whole batch.
cfg3's shape with longer IRs: C = 32 capsules, E = 16 events x N = 16 waypoint IRs, 7.75 s clips @ 48 kHz
(python3 quad16_moving_sweep.py [C E N], LIRS="..." for the IR lengths)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from audiblelight_amd import engine, plan as planning

r = engine.Renderer()
rng = np.random.default_rng(0)
C, E, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 16, 16)))
La, sr = 372000, 48000
clips = [rng.standard_normal(La).astype(np.float32) for _ in range(E)]
specs = [planning.EventSpec(n_samples=La, n_emitters=N, snr=10.0, emitter0=e * N, is_moving=True, duration=La / sr) for e in range(E)]
for Lir in [int(x) for x in os.environ.get("LIRS", "110000 150000 192000").split()]:
    irs = torch.randn((C * E * N, Lir), device="cuda", dtype=torch.float32)
    irs *= torch.exp(-torch.arange(Lir, device="cuda", dtype=torch.float32) / (Lir / 6.9))[None, :]
    irs = irs.reshape(-1)
    row = []
    for lb in (13, 14):
        pl = planning.plan_batch(specs, C, Lir, sr, log2_block=lb)
        batch = r.prepare(pl, clips, irs, ir_strides=(E * N * Lir, Lir))
        import ctypes as ct
        s_code, m_code = ct.c_int32(), ct.c_int32()
        r.lib.call("al_spectral_mac_variant", ct.byref(batch.descs[0]), ct.byref(s_code), ct.byref(m_code))
        for _ in range(2):
            batch.run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(4):
                batch.run()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 4 * 1e3)
        row.append((lb, pl.n_partitions, m_code.value, best))
        del batch
        torch.cuda.empty_cache()
    print(f"Lir={Lir}: " + "  ".join(f"B=2^{lb} P={p} moving code {m}: {t:.3f} ms" for lb, p, m, t in row) + f"  ratio {row[1][3] / row[0][3]:.3f}", flush=True)
    del irs
