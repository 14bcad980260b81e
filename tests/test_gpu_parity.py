"""Parity of the gfx950 build (through the C ABI) against the reference goldens and the oracle.

Tolerance (BASELINE.json north_star): relative RMS <= 1e-4 and max-abs error <= 1e-4 * max|ref|
against the float64 reference cast to float32 (SURVEY.md 8d "Parity metric").
"""
import numpy as np
import pytest

from oracle import synth_oracle as orc
from tests.conftest import assert_parity, rel_rms, set_switch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def assert_close(got, ref, tol=TOL):
    """The contract's bound, both halves: relative RMS <= tol and max|err| <= tol * max|ref|."""
    assert_parity(got, ref, tol)


@pytest.fixture(scope="module")
def gpu():
    from audiblelight_amd import engine

    r = engine.Renderer()  # raises if the HIP extension or the GPU is missing: no fallback
    assert r.lib.path.endswith("libaudiblelight_hip.so")
    return r


@pytest.fixture(scope="module")
def planning():
    from audiblelight_amd import plan

    return plan


@pytest.mark.parametrize("log2_block", [10, 11, 12, 13, 14])
def test_static_golden(gpu, planning, golden, log2_block):
    a, h = golden["g1_audio"], golden["g1_irs"]
    pl = planning.plan_batch([planning.EventSpec(n_samples=len(a), n_emitters=1, snr=10.0)], 4, h.shape[2], 8000,
                             log2_block=log2_block)
    res = gpu.render(pl, [a], h)
    assert_close(res.spatial_audio(0), golden["g1_spatial"])
    np.testing.assert_allclose(gpu.mem.download(res.emitter_gain)[:1], orc.emitter_gains(h.astype(np.float64)), rtol=1e-5)
    res.check_finite()


def test_static_ir_longer_than_clip(gpu, planning, golden):
    a, h = golden["g1b_audio"], golden["g1b_irs"]
    pl = planning.plan_batch([planning.EventSpec(n_samples=len(a), n_emitters=1, snr=0.5, ref_db=-50)], 3, h.shape[2], 8000)
    assert_close(gpu.render(pl, [a], h).spatial_audio(0), golden["g1b_spatial"])


def test_zero_emitter(gpu, planning, golden):
    a = golden["g2_audio"]
    pl = planning.plan_batch([planning.EventSpec(n_samples=len(a), n_emitters=0, snr=7.0)], 4, 100, 8000, log2_block=10)
    assert_close(gpu.render(pl, [a], np.zeros((4, 0, 100))).spatial_audio(0), golden["g2_spatial"], 1e-6)


@pytest.mark.parametrize("tag,n_ir", [("g3a", 3), ("g3b", 5)])
@pytest.mark.parametrize("log2_block", [10, 12])
def test_moving_golden(gpu, planning, golden, tag, n_ir, log2_block):
    a, h = golden[f"{tag}_audio"], golden[f"{tag}_irs"]
    spec = planning.EventSpec(n_samples=len(a), n_emitters=n_ir, snr=12.0, is_moving=True, duration=len(a) / 8000)
    pl = planning.plan_batch([spec], 4, h.shape[2], 8000, log2_block=log2_block)
    assert_close(gpu.render(pl, [a], h).spatial_audio(0), golden[f"{tag}_spatial"])


def test_full_scene_golden(gpu, planning, golden):
    sr, dur, C = 8000, 2.0, 4
    specs, clips, irs, col = [], [], [], 0
    for i, (na, ne, st, snr, mv, dry) in enumerate(golden["g8_specs"]):
        a = golden[f"g8_audio{i}"]
        clips.append(a)
        irs.append(golden[f"g8_irs{i}"])
        specs.append(planning.EventSpec(n_samples=len(a), n_emitters=int(ne), snr=float(snr), emitter0=col,
                                        is_moving=bool(mv), duration=len(a) / sr))
        col += int(ne)
    pl = planning.plan_batch(specs, C, 1000, sr, log2_block=10)
    res = gpu.render(pl, clips, np.concatenate(irs, axis=1))
    for i in range(len(specs)):
        assert_close(res.spatial_audio(i), golden[f"g8_spatial{i}"])
    starts = [float(s[2]) for s in golden["g8_specs"]]
    mix = planning.plan_mixdown(starts, [s + len(c) / sr for s, c in zip(starts, clips)], [len(c) for c in clips],
                                [C] * 5, pl.events["out_off"], list(range(5)), dur, sr, C)
    amb = golden["g8_ambience"].astype(np.float32)
    amb_dev = gpu.mem.upload(amb.reshape(-1))
    stats = gpu.mem.download(gpu.row_stats(amb_dev, 1, amb.size)).reshape(-1, 4)
    assert stats[0, 0] == pytest.approx(np.abs(amb.astype(np.float64)).sum(), rel=1e-6)
    mult = np.float32(orc.db_gain(-65, stats[0, 0] / amb.size))
    scene = gpu.mem.download(gpu.mixdown(mix, res, [(amb_dev, gpu.mem.upload(np.full(mix.n_capsules, mult, np.float32)))]))
    assert_close(scene[: C * mix.n_samples].reshape(C, -1), golden["g8_scene"])


@pytest.mark.parametrize("n_audio,n_ir,C", [(1, 7, 3), (777, 1, 1), (5001, 3333, 5), (1024, 1024, 2), (1025, 2049, 3),
                                            (20000, 300, 4)])
def test_ragged_edges_vs_oracle(gpu, planning, n_audio, n_ir, C):
    rng = np.random.default_rng(n_audio + n_ir)
    a = rng.standard_normal(n_audio).astype(np.float32)
    h = (rng.standard_normal((C, 1, n_ir)) * np.exp(-np.arange(n_ir) / max(n_ir / 5, 1))).astype(np.float32)
    pl = planning.plan_batch([planning.EventSpec(n_samples=n_audio, n_emitters=1, snr=11.0)], C, n_ir, 16000, log2_block=10)
    got = gpu.render(pl, [a], h).spatial_audio(0)
    want = orc.render_event(a, h.astype(np.float64), 11.0, sr=16000)["spatial"]
    assert_close(got, want)


def test_cfg1_scene_vs_oracle(gpu, planning):
    """BASELINE configs[0]: 1 scene, 4 capsules, 4 static events, 0.5 s RIRs @ 24 kHz."""
    from audiblelight_amd import synthetic

    sc = synthetic.make_scene("cfg1")
    pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
    res = gpu.render(pl, sc.clips, sc.irs)
    want = [orc.render_event(a, sc.irs[:, [i], :].astype(np.float64), sp.snr, sr=sc.sr)["spatial"]
            for i, (a, sp) in enumerate(zip(sc.clips, sc.specs))]
    for i, w in enumerate(want):
        assert_close(res.spatial_audio(i), w)
    mix = planning.plan_mixdown(sc.starts, sc.ends, [len(c) for c in sc.clips], [sc.n_capsules] * 4,
                                pl.events["out_off"], list(range(4)), sc.duration, sc.sr, sc.n_capsules)
    scene = gpu.mem.download(gpu.mixdown(mix, res))[: sc.n_capsules * mix.n_samples].reshape(sc.n_capsules, -1)
    ref = orc.mix_scene(want, list(zip(sc.starts, sc.ends)), sc.duration, sc.sr, keep_padded=False)["scene"]
    assert_close(scene, ref)


def test_mixed_batch_moving_and_static_vs_oracle(gpu, planning):
    rng = np.random.default_rng(5)
    sr, C, L = 16000, 6, 5000
    specs, clips, irs, col = [], [], [], 0
    for n_audio, n_emit in ((30000, 1), (41000, 4), (12345, 1), (25000, 7), (9000, 0)):
        a = rng.standard_normal(n_audio).astype(np.float32)
        clips.append(a / np.abs(a).max())
        irs.append((rng.standard_normal((C, n_emit, L)) * np.exp(-np.arange(L) / 700.0)).astype(np.float32))
        specs.append(planning.EventSpec(n_samples=n_audio, n_emitters=n_emit, snr=float(rng.uniform(5, 30)), emitter0=col,
                                        is_moving=n_emit > 1, duration=n_audio / sr))
        col += n_emit
    pl = planning.plan_batch(specs, C, L, sr, log2_block=12)
    res = gpu.render(pl, clips, np.concatenate(irs, axis=1))
    for i, (a, h, sp) in enumerate(zip(clips, irs, specs)):
        want = orc.render_event(a, h.astype(np.float64), sp.snr, is_moving=sp.is_moving, duration=sp.duration, sr=sr)
        assert_close(res.spatial_audio(i), want["spatial"])


def test_nonfinite_input_is_flagged(gpu, planning):
    a = np.ones(3000, dtype=np.float32)
    a[100] = np.inf
    h = np.ones((2, 1, 50), dtype=np.float32)
    pl = planning.plan_batch([planning.EventSpec(n_samples=3000, n_emitters=1, snr=5.0)], 2, 50, 8000, log2_block=10)
    with pytest.raises(ValueError, match="not finite"):
        gpu.render(pl, [a], h).check_finite()


@pytest.mark.parametrize("name", ["cfg2"])
def test_full_size_properties(gpu, planning, name):
    """BASELINE configs[1] at full size: size-independent properties + spot rows against the oracle."""
    from audiblelight_amd import synthetic

    sc = synthetic.make_scene(name)
    # event 0: unit impulse IRs -> the render is the delayed clip (identity property)
    delays = np.arange(sc.n_capsules) * 37 + 5
    sc.irs[:, 0, :] = 0
    sc.irs[np.arange(sc.n_capsules), 0, delays] = 1.0
    pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
    res = gpu.render(pl, sc.clips, sc.irs)
    res.check_finite()
    scales, stats = res.scales(), res.stats()
    # A9 invariant: mean|out| == 10^((ref_db + snr)/20) for every event
    for e, sp in enumerate(sc.specs):
        mean_abs = scales[e] * stats[e, 0] / (sc.n_capsules * sp.n_samples)
        assert mean_abs == pytest.approx(10 ** ((sp.ref_db + sp.snr) / 20), rel=1e-5)
    raw0 = res.raw_spatial(0)
    g0 = float(gpu.mem.download(res.emitter_gain)[0])
    assert g0 == pytest.approx(1.0, rel=1e-6)
    for c in (0, 7, 31):
        want = np.zeros(sc.specs[0].n_samples)
        want[delays[c]:] = sc.clips[0][: sc.specs[0].n_samples - delays[c]]
        assert rel_rms(raw0[c], want) < 1e-5
    # spot rows against the float64 oracle at full size
    from scipy.signal import fftconvolve
    gains = gpu.mem.download(res.emitter_gain)
    for e, c in ((3, 5), (40, 31), (63, 0)):
        ref = fftconvolve(sc.clips[e].astype(np.float64), sc.irs[c, e].astype(np.float64))[: sc.specs[e].n_samples]
        assert_close(res.raw_spatial(e)[c], ref * gains[e])
        e_ref = orc.emitter_gains(sc.irs[:, [e], :].astype(np.float64))[0]
        assert gains[e] == pytest.approx(e_ref, rel=1e-5)
    # mixdown of the whole scene: every output sample is the float32 sum the kernel's own inputs imply (all 64 events)
    mix = planning.plan_mixdown(sc.starts, sc.ends, [len(c) for c in sc.clips], [sc.n_capsules] * len(sc.clips),
                                pl.events["out_off"], list(range(len(sc.clips))), sc.duration, sc.sr, sc.n_capsules)
    scene = gpu.mem.download(gpu.mixdown(mix, res))[: sc.n_capsules * mix.n_samples].reshape(sc.n_capsules, -1)
    c = 9
    want = np.zeros(mix.n_samples)
    for e in range(len(sc.clips)):
        a0, b0 = planning.event_slot(sc.starts[e], sc.ends[e], sc.sr, mix.n_samples)
        want[a0:b0] += (res.raw_spatial(e)[c].astype(np.float64) * scales[e])[: b0 - a0]
    assert rel_rms(scene[c], want) < 1e-6
    # ... and against the ORACLE's scene at full length (60 s, all 32 capsules, every row) for the first six events: the
    # whole chain render -> level law -> mixdown, nothing taken from the device (the oracle needs 0.3 s per full-size event)
    n6 = 6
    pl6 = planning.plan_batch(sc.specs[:n6], sc.n_capsules, sc.ir_len, sc.sr)
    res6 = gpu.render(pl6, sc.clips[:n6], sc.irs[:, :n6, :])
    mix6 = planning.plan_mixdown(sc.starts[:n6], sc.ends[:n6], [len(c) for c in sc.clips[:n6]], [sc.n_capsules] * n6,
                                 pl6.events["out_off"], list(range(n6)), sc.duration, sc.sr, sc.n_capsules)
    scene6 = gpu.mem.download(gpu.mixdown(mix6, res6))[: sc.n_capsules * mix6.n_samples].reshape(sc.n_capsules, -1)
    spat = [orc.render_event(sc.clips[e], sc.irs[:, [e], :].astype(np.float64), sc.specs[e].snr, sc.specs[e].ref_db, sr=sc.sr)["spatial"]
            for e in range(n6)]
    ref6 = orc.mix_scene(spat, list(zip(sc.starts[:n6], sc.ends[:n6])), sc.duration, sc.sr, keep_padded=False)["scene"]
    assert scene6.shape == ref6.shape == (sc.n_capsules, 2880000)
    for row in range(sc.n_capsules):
        assert_parity(scene6[row], ref6[row], TOL, what=row)


def test_cfg2_full_scene_vs_oracle(gpu, planning):
    """The headline workload in full against the oracle: BASELINE configs[1] -- 64 events x 32 capsules x 2 s RIRs, 60 s at
    48 kHz -- rendered and mixed through the default dispatch (the configuration bench.py times) and compared with the oracle's
    scene of ALL 64 events, every capsule row x all 2 880 000 samples, both halves of the 1e-4 bound; every event's level-law
    scalar against the oracle's as well.  (bench.py makes the same comparison on the buffer its timed steps wrote: JSON key
    `parity`.)  Reference: synthesize.py:613-677 then :314-401."""
    from audiblelight_amd import synthetic

    sc = synthetic.make_scene("cfg2")
    pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr, lib=gpu.lib)
    batch = gpu.prepare(pl, sc.clips, sc.irs)
    import ctypes as ct
    code, mcode = ct.c_int32(), ct.c_int32()
    gpu.lib.call("al_spectral_mac_variant", ct.byref(batch.descs[0]), ct.byref(code), ct.byref(mcode))
    assert (pl.log2_block, code.value) == (13, 3121202)          # split layout, k_spectral_mac_static<12,12,2>
    res = batch.run()
    res.check_finite()
    n = len(sc.clips)
    mix = planning.plan_mixdown(sc.starts, sc.ends, [len(c) for c in sc.clips], [sc.n_capsules] * n, pl.events["out_off"],
                                list(range(n)), sc.duration, sc.sr, sc.n_capsules, lib=gpu.lib)
    scene = gpu.mem.download(gpu.mixdown(mix, res))[: sc.n_capsules * mix.n_samples].reshape(sc.n_capsules, -1)
    scales = res.scales()
    ref = np.zeros((sc.n_capsules, mix.n_samples), dtype=np.float32)
    for e in range(n):          # one event at a time: the oracle's float64 (32, 192 000) render is 49 MB
        out = orc.render_event(sc.clips[e], sc.irs[:, [e], :].astype(np.float64), sc.specs[e].snr, sc.specs[e].ref_db, sr=sc.sr)
        a0, b0 = orc.event_slot(sc.starts[e], sc.ends[e], sc.sr, mix.n_samples)
        ref[:, a0:b0] += orc.fit_length(out["spatial"], b0 - a0)
        # total multiplier of the raw convolution: unit_energy gain cancels in the level law for static events
        assert scales[e] * res.stats()[e, 0] / (sc.n_capsules * sc.specs[e].n_samples) == pytest.approx(
            10 ** ((sc.specs[e].ref_db + sc.specs[e].snr) / 20), rel=1e-5)
    assert scene.shape == ref.shape == (32, 2_880_000)
    assert_parity(scene, ref, TOL, what="cfg2 scene, 64 events")
    for row in range(sc.n_capsules):
        assert_parity(scene[row], ref[row], TOL, what=row)


@pytest.mark.parametrize("seed", range(12))
def test_random_batches_vs_oracle(gpu, planning, seed, monkeypatch):
    """Seeded random batches (static / moving / zero-emitter events, ragged lengths, every block size, runs of blocks
    per workgroup on or off, odd capsule counts) against the float64 oracle."""
    rng = np.random.default_rng(1000 + seed)
    log2_block = int(rng.integers(10, 15))
    if seed % 3 == 1:
        set_switch(monkeypatch, "AL_EXTRA_FLAGS", str((int(rng.integers(2, 5)) << 16) | (int(rng.integers(2, 5)) << 24)))
    elif seed % 3 == 2:
        set_switch(monkeypatch, "AL_EXTRA_FLAGS", "4")
    sr, C = 16000, int(rng.integers(1, 8))
    L = int(rng.integers(1, 3 << log2_block))
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
    pl = planning.plan_batch(specs, C, L, sr, log2_block=log2_block)
    res = gpu.render(pl, clips, np.concatenate(irs, axis=1))
    res.check_finite()
    for i, (a, h, sp) in enumerate(zip(clips, irs, specs)):
        want = orc.render_event(a, h.astype(np.float64), sp.snr, is_moving=sp.is_moving, duration=sp.duration, sr=sr)
        assert_close(res.spatial_audio(i), want["spatial"])


def test_static_vs_direct_form_c_witness(gpu, planning):
    """The HIP convolution against oracle/conv_direct.c: a time-domain float64 sum in plain C that shares no FFT, no
    numpy and no code with the other oracle (ragged lengths, IR longer than one block, 3 capsules)."""
    from oracle import conv_direct as cd

    rng = np.random.default_rng(77)
    a = rng.standard_normal(3001).astype(np.float32)
    h = (rng.standard_normal((3, 1, 2500)) * np.exp(-np.arange(2500) / 400.0)).astype(np.float32)
    pl = planning.plan_batch([planning.EventSpec(n_samples=len(a), n_emitters=1, snr=9.0)], 3, 2500, 16000, log2_block=10)
    res = gpu.render(pl, [a], h)
    gain = orc.emitter_gains(h.astype(np.float64))[0]
    want = cd.conv_direct(a, h[:, 0].astype(np.float64) * gain, len(a))
    assert_close(res.raw_spatial(0), want)
