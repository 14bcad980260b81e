"""Per-stage time and effective HBM rate (designed bytes / time) over shapes around cfg2, one parameter at a time: looks for
performance cliffs away from the headline shape.  Static events only."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from audiblelight_amd import engine, plan as planning
r = engine.Renderer()
rng = np.random.default_rng(0)
base = dict(C=32, E=64, La=192000, Lir=96000)
sweeps = [("base", {})] + [("La", dict(La=v)) for v in (48000, 100000, 160000, 400000, 960000)] + \
         [("Lir", dict(Lir=v)) for v in (4000, 30000, 100000, 110000, 150000, 192000)] + \
         [("C", dict(C=v)) for v in (1, 4, 19, 64)] + [("E", dict(E=v)) for v in (1, 4, 16)] + \
         [("ragged", dict(ragged=True)), ("odd", dict(La=191999)), ("odd", dict(La=191997, Lir=95999))]
for name, over in sweeps:
    p = dict(base, **over)
    C, E, La, Lir = p["C"], p["E"], p["La"], p["Lir"]
    if C * E * Lir > 32 * 64 * 192000:
        E = max(1, 32 * 64 * 192000 // (C * Lir))
    lens = [La - (137 * e if p.get("ragged") else 0) - (e * 9973 % 50000 if p.get("ragged") else 0) for e in range(E)]
    clips = [rng.standard_normal(n).astype(np.float32) for n in lens]
    irs = (rng.standard_normal((C, E, Lir), dtype=np.float32) * np.exp(-np.arange(Lir) / (Lir / 6.9)).astype(np.float32))
    specs = [planning.EventSpec(n_samples=n, n_emitters=1, snr=10.0, emitter0=e) for e, n in enumerate(lens)]
    pl = planning.plan_batch(specs, C, Lir, 48000)
    batch = r.prepare(pl, clips, irs)
    for _ in range(3): batch.run()
    torch.cuda.synchronize()
    B, P = pl.block, pl.n_partitions
    Ksum = int(pl.events["n_blocks"].sum())
    hb, yb, xb = E * C * P * B * 8, C * Ksum * B * 8, Ksum * B * 8
    designed = {"al_forward_spectra": C * E * Lir * 4 + hb + sum(lens) * 4 + xb, "al_spectral_mac": hb + yb + xb,
                "al_block_synthesis": yb + C * sum(lens) * 4}
    tot = {}
    reps = 10
    for _ in range(reps):
        for st in batch.stage_names(0):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r.lib.call(st, ctypes.byref(batch.descs[0]), r.mem.stream()); b.record()
            torch.cuda.synchronize()
            tot[st] = tot.get(st, 0.0) + a.elapsed_time(b) / reps
    s_code = ctypes.c_int32(); m_code = ctypes.c_int32()
    r.lib.call("al_spectral_mac_variant", ctypes.byref(batch.descs[0]), ctypes.byref(s_code), ctypes.byref(m_code))
    line = " ".join(f"{k[3:]} {tot[k]:.3f} ms {designed[k] / tot[k] / 1e9:5.2f} TB/s |" for k in designed)
    print(f"{name:6s} C={C:2d} E={E:2d} La={La:6d} Lir={Lir:6d} lb={pl.log2_block} P={P:2d} K={int(pl.events['n_blocks'].max()):3d} mac={s_code.value} | {line}", flush=True)
    del batch, irs, clips
    torch.cuda.empty_cache()
