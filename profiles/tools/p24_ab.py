"""A/B of the accumulate for 17..24 partitions: capsule-loop kernel with three units per capsule (k_spectral_mac_static_lds<12,PT,3>)
against the tile kernels (AL_STATIC_MAC_MAX_P=16 keeps the old dispatch).  Static events, C=32, E=64, La=192000."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
import numpy as np, torch
from audiblelight_amd import engine, plan as planning
r = engine.Renderer()
rng = np.random.default_rng(0)
for Lir in (100000, 125000, 135000, 150000, 170000, 192000):
    for max_p, flags in (("12", "0"), ("18", "0"), ("24", "16384")):    # tile kernels | register / LDS-staged capsule loop | LDS-DMA capsule loop
        if max_p == "18" and Lir > 147456:
            continue
        _sw.set_env("AL_STATIC_MAC_MAX_P", max_p); _sw.set_env("AL_EXTRA_FLAGS", flags)
        C, E, La = 32, 64, 192000
        clips = [rng.standard_normal(La).astype(np.float32) for _ in range(E)]
        irs = (rng.standard_normal((C, E, Lir), dtype=np.float32) * np.exp(-np.arange(Lir) / (Lir / 6.9)).astype(np.float32))
        specs = [planning.EventSpec(n_samples=La, n_emitters=1, snr=10.0, emitter0=e) for e in range(E)]
        pl = planning.plan_batch(specs, C, Lir, 48000)
        batch = r.prepare(pl, clips, irs)
        for _ in range(3): batch.run()
        torch.cuda.synchronize()
        B, P = pl.block, pl.n_partitions
        Ksum = int(pl.events["n_blocks"].sum())
        designed = E * C * P * B * 8 + C * Ksum * B * 8 + Ksum * B * 8
        t = 0.0
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r.lib.call("al_spectral_mac", ctypes.byref(batch.descs[0]), r.mem.stream()); b.record()
            torch.cuda.synchronize()
            t += a.elapsed_time(b) / 10
        s_code = ctypes.c_int32(); m_code = ctypes.c_int32()
        r.lib.call("al_spectral_mac_variant", ctypes.byref(batch.descs[0]), ctypes.byref(s_code), ctypes.byref(m_code))
        print(f"Lir={Lir} P={P} mac={s_code.value}: {t:.3f} ms  {designed / t / 1e9:.2f} TB/s", flush=True)
        del batch, irs, clips
        torch.cuda.empty_cache()
