"""Lease-side evidence for k_moving_fused (csrc/al_quad.h, AL_FUSED_MOVING=1): seeded random batches of sliding-window moving events
at B = 8192 -- 1..24 partitions with ragged tails, 4..40 IRs, cross-fade windows of 2..6 blocks (both instantiations), clips that cut
the late IRs' partitions off, sometimes a static or a tiled event in the same batch (quad layout through the other kernels) --
every row against the float64 oracle with the contract's bound, on W worker processes that share the GPU.
    AL_FUSED_MOVING=1 python3 profiles/tools/fuzz_fused_moving.py FIRST LAST [WORKERS]"""
import collections, os, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)


def one(r, seed):
    import numpy as np
    from audiblelight_amd import plan as planning
    from oracle import synth_oracle as orc
    from tests import mac_regimes as mr

    rng = np.random.default_rng(70_000 + seed)
    B, sr = 8192, 48000
    C = int(rng.integers(1, 5))
    Lir = int(rng.integers(1, 24 * B))
    specs, clips, irs, col = [], [], [], 0
    for _ in range(int(rng.integers(1, 3))):
        n_irs = int(rng.integers(4, 41))
        seg = float(rng.uniform(0.7, 2.0))                       # blocks between consecutive IRs: windows of 2.4 .. 6 blocks
        n = max(int(seg * (n_irs - 1) * B + rng.integers(-3000, 3000)), 2 * B)
        a = rng.standard_normal(n).astype(np.float32)
        clips.append(a / np.abs(a).max())
        irs.append((rng.standard_normal((C, n_irs, Lir)) * np.exp(-np.arange(Lir) / max(Lir / 5.0, 1.0))).astype(np.float32))
        specs.append(planning.EventSpec(n_samples=n, n_emitters=n_irs, snr=float(rng.uniform(5, 30)), emitter0=col, is_moving=True, duration=n / sr))
        col += n_irs
    if rng.random() < 0.4:                                       # a static or tiled event beside them
        ne = int(rng.integers(0, 2))
        n = int(rng.integers(1, 20 * B))
        a = rng.standard_normal(n).astype(np.float32)
        clips.append(a / np.abs(a).max())
        irs.append((rng.standard_normal((C, ne, Lir)) * np.exp(-np.arange(Lir) / max(Lir / 5.0, 1.0))).astype(np.float32))
        specs.append(planning.EventSpec(n_samples=n, n_emitters=ne, snr=float(rng.uniform(5, 30)), emitter0=col))
        col += ne
    pl = planning.plan_batch(specs, C, Lir, sr, log2_block=13)
    batch = r.prepare(pl, clips, np.concatenate(irs, axis=1))
    codes = mr.mac_codes(r, batch)
    res = batch.run()
    res.check_finite()
    for i, (a, h, sp) in enumerate(zip(clips, irs, specs)):
        want = orc.render_event(a, h.astype(np.float64), sp.snr, is_moving=sp.is_moving, duration=sp.duration, sr=sr)["spatial"]
        mr.check_event_rows(res, i, want)
    return codes[1], pl.n_partitions, int(pl.streams["n_j"].max())


def work(span):
    from audiblelight_amd import engine
    r = engine.Renderer()
    out = []
    for seed in range(*span):
        try:
            out.append((seed,) + one(r, seed) + (None,))
        except Exception as exc:  # noqa: BLE001 -- a fuzz driver reports everything
            out.append((seed, -1, -1, -1, f"{type(exc).__name__}: {str(exc)[:300]}"))
    return out


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    step = max(1, (last - first + workers * 4 - 1) // (workers * 4))
    spans = [(a, min(last, a + step)) for a in range(first, last, step)]
    t0 = time.time()
    with ProcessPoolExecutor(workers) as pool:
        rows = [row for part in pool.map(work, spans) for row in part]
    bad = [row for row in rows if row[4]]
    print(f"AL_FUSED_MOVING={os.environ.get('AL_FUSED_MOVING', '0')}: seeds {first}..{last - 1}: {len(rows)} batches, {len(bad)} failed, {time.time() - t0:.0f} s on {workers} processes")
    print("moving-accumulate code -> batches:", dict(collections.Counter(row[1] for row in rows)))
    print("partitions seen:", sorted({row[2] for row in rows if row[2] > 0}))
    print("longest stream (blocks) -> batches:", dict(collections.Counter(row[3] for row in rows)))
    for row in bad:
        print("FAILED", row)
    sys.exit(1 if bad else 0)
