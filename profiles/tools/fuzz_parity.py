import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
from audiblelight_amd import engine, plan as planning
from oracle import synth_oracle as orc
r = engine.Renderer()
bad = 0
checked = 0
worst = 0.0
for seed in range(int(os.environ.get("FUZZ_FROM", 0)), int(os.environ.get("FUZZ_TO", 400))):
    rng = np.random.default_rng(5000 + seed)
    log2_block = int(rng.integers(10, 15))
    m = seed % 3
    _sw.set_env("AL_EXTRA_FLAGS", str((int(rng.integers(2, 5)) << 16) | (int(rng.integers(2, 5)) << 24)) if m == 1 else ("4" if m == 2 else "0"))
    sr, C = 16000, int(rng.integers(1, 8))
    if seed >= 400:   # long IRs (up to 14 partitions: both accumulate families) at small blocks, transform layout and accumulate kernel at random
        log2_block = int(rng.integers(10, 13))
        _sw.set_env("AL_SPLIT", str(int(rng.integers(0, 2))))
        _sw.set_env("AL_STATIC_MAC", str(int(rng.integers(0, 2))))
    L = int(rng.integers(1, (14 if seed >= 400 else 3) << log2_block))
    specs, clips, irs, col = [], [], [], 0
    for _ in range(int(rng.integers(1, 5))):
        kind = rng.choice(["static", "static", "moving", "dry"])
        n_audio = int(rng.integers(600, 5 << log2_block)) if kind == "moving" else int(rng.integers(1, 5 << log2_block))
        n_emit = {"static": 1, "dry": 0, "moving": int(rng.integers(2, 7))}[kind]
        a = rng.standard_normal(n_audio).astype(np.float32)
        clips.append(a / np.abs(a).max())
        irs.append((rng.standard_normal((C, n_emit, L)) * np.exp(-np.arange(L) / max(L / 6, 1))).astype(np.float32))
        specs.append(planning.EventSpec(n_samples=n_audio, n_emitters=n_emit, snr=float(rng.uniform(5, 30)), emitter0=col,
                                        is_moving=n_emit > 1, duration=n_audio / sr))
        col += n_emit
    try:
        pl = planning.plan_batch(specs, C, L, sr, log2_block=log2_block)
        res = r.render(pl, clips, np.concatenate(irs, axis=1))
        res.check_finite()
        for i, (a, h, sp) in enumerate(zip(clips, irs, specs)):
            want = orc.render_event(a, h.astype(np.float64), sp.snr, is_moving=sp.is_moving, duration=sp.duration, sr=sr)["spatial"]
            got = res.spatial_audio(i)
            den = np.sqrt(np.mean(want ** 2))
            err = np.sqrt(np.mean((got - want) ** 2)) / den if den > 0 else float(np.abs(got).max())
            worst = max(worst, err)
            checked += 1
            if not err < 1e-4:
                bad += 1
                print("FAIL seed", seed, "event", i, "err", err, "lb", log2_block, "C", C, "L", L, len(a), sp.n_emitters, os.environ["AL_EXTRA_FLAGS"], flush=True)
    except Exception as ex:
        bad += 1
        print("EXC seed", seed, type(ex).__name__, str(ex)[:200], "lb", log2_block, "C", C, "L", L, [(len(c), s.n_emitters) for c, s in zip(clips, specs)], flush=True)
print("done: events checked", checked, "failures", bad, "worst rel rms", worst)
