"""BASELINE.json configs[2..4] at FULL size on the MI355X (the sizes bench.py times), shaped like
tests/test_gpu_parity.py::test_full_size_properties for configs[1]: size-independent properties on every event (finite,
the A9 level invariant mean|out| = 10^((ref_db + snr)/20)), spot rows against the float64 oracle, and one
scene row against the float32 sum the kernel's own inputs imply.  The oracle cannot render these scenes in full (cfg3: hours,
cfg5: 94 GB of padded copies), so it renders the rows it is asked for from the same inputs.

Reference: audiblelight/synthesize.py:184-310 (moving), :71-106 (static), :314-401 (mixdown), :594-599 (level law).
"""
import numpy as np
import pytest
from scipy.signal import fftconvolve

from oracle import synth_oracle as orc
from tests.conftest import assert_parity, rel_rms

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def gpu():
    from audiblelight_amd import engine

    r = engine.Renderer()
    assert r.lib.path.endswith("libaudiblelight_hip.so")
    return r


def row_of(res, e, c):
    """One capsule row of event e's UNSCALED render, downloaded alone (an event is 25-50 MB, a scene of them gigabytes)."""
    ev = res.plan.events[e]
    n = int(ev["len"])
    off = int(ev["out_off"]) + c * n
    return np.asarray(res.memory.download(res.spatial[off: off + n]), dtype=np.float64)


def check_every_event_against_the_oracle(sc, res):
    """EVERY event of the full-size render meets the float64 oracle on one pseudo-random capsule row (bench.oracle_row_samples: the
    check the bench line's `parity.rows_sampled` carries), on top of the hand-picked spot rows of each test."""
    import bench

    cols = lambda e: slice(sc.specs[e].emitter0, sc.specs[e].emitter0 + sc.specs[e].n_emitters)    # noqa: E731
    rec = bench.oracle_row_samples(sc, res, range(len(sc.specs)), lambda e, c: sc.irs[c, cols(e), :],
                                   lambda e: orc.emitter_gains(sc.irs[:, cols(e), :]))
    assert rec["events"] == len(sc.specs) and rec["ok"], {k: v for k, v in rec.items() if k != "note"}
    assert rec["rel_rms_worst_row"] <= TOL and rec["max_abs_over_peak_worst_row"] <= TOL
    return rec


def check_level_invariant(sc, res):
    """A9 on EVERY event: mean|scale * x| over (C, La) equals 10^((ref_db + snr)/20); and nothing is non-finite."""
    res.check_finite()
    scales, stats = res.scales(), res.stats()
    for e, sp in enumerate(sc.specs):
        mean_abs = scales[e] * stats[e, 0] / (sc.n_capsules * sp.n_samples)
        assert mean_abs == pytest.approx(10 ** ((sp.ref_db + sp.snr) / 20), rel=1e-5), e
    return scales


def check_scene_row(gpu, planning, sc, pl, res, scales, c, ambience=()):
    """Row c of the mixdown against the sum the kernel's inputs imply: ambience row + sum_e scale_e * x_e[c] at its slot."""
    n_ev = len(sc.clips)
    mix = planning.plan_mixdown(sc.starts, sc.ends, [len(a) for a in sc.clips], [sc.n_capsules] * n_ev, pl.events["out_off"],
                                list(range(n_ev)), sc.duration, sc.sr, sc.n_capsules)
    scene_dev = gpu.mixdown(mix, res, list(ambience))
    n = mix.n_samples
    got = np.asarray(gpu.mem.download(scene_dev[c * n: (c + 1) * n]))
    want = np.zeros(n, dtype=np.float32)
    for noise, amb_scales in ambience:
        a = np.asarray(gpu.mem.download(noise[c * n: (c + 1) * n]), dtype=np.float32)
        want += np.float32(np.asarray(gpu.mem.download(amb_scales))[c]) * a
    for e in range(n_ev):
        a0, b0 = planning.event_slot(sc.starts[e], sc.ends[e], sc.sr, n)
        if b0 > a0:
            want[a0:b0] += (np.float32(scales[e]) * row_of(res, e, c).astype(np.float32))[: b0 - a0]
    assert np.isfinite(got).all()
    assert rel_rms(got, want) < 1e-6
    return got


_CFG3 = {}


def _cfg3_scene():
    """The full-size cfg3 scene, drawn once for both parametrisations (6.3 GB of host draws take 20 s)."""
    from audiblelight_amd import synthetic

    if "scene" not in _CFG3:
        _CFG3["scene"] = synthetic.make_scene("cfg3")
    return _CFG3["scene"]


def test_cfg3_full_size_moving_sources(gpu):
    """BASELINE configs[2]: 16 moving events x 32 waypoint IRs, 32 capsules, 2 s RIRs, 7.75 s clips @ 48 kHz (6.3 GB of IRs)
    through the default dispatch (sliding-window accumulate over stored IR spectra)."""
    from audiblelight_amd import plan as planning
    from tests import mac_regimes as mr

    sc = _cfg3_scene()
    assert sc.irs.shape == (32, 16 * 32, 96000) and len(sc.clips) == 16 and len(sc.clips[0]) == 372000
    pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
    assert pl.log2_block == 13 and pl.n_partitions == 12 and int(pl.events["n_blocks"].max()) == 46
    assert all(int(r) == 1 for r in pl.events["reserved"])
    batch = gpu.prepare(pl, sc.clips, sc.irs)
    moving_code = mr.mac_codes(gpu, batch)[1]
    assert moving_code == 612, moving_code
    res = batch.run()
    scales = check_level_invariant(sc, res)
    gains = np.asarray(gpu.mem.download(res.emitter_gain))[: 16 * 32].astype(np.float64)
    for e, caps in ((0, (3, 17)), (9, (31,)), (15, (0,))):
        sp = sc.specs[e]
        h = sc.irs[:, sp.emitter0: sp.emitter0 + 32, :]
        g_ref = orc.emitter_gains(h)                                     # over ALL 32 capsules (synthesize.py:404-428)
        np.testing.assert_allclose(gains[sp.emitter0: sp.emitter0 + 32], g_ref, rtol=2e-5)
        sel = h[list(caps)].astype(np.float64) * g_ref[None, :, None]
        want = orc.fit_length(orc.convolve_moving(sc.clips[e], sel, sp.duration, sc.sr), sp.n_samples)
        for i, c in enumerate(caps):
            assert_parity(row_of(res, e, c), want[i], TOL, what=(e, c))
    assert check_every_event_against_the_oracle(sc, res)["events"] == 16
    check_scene_row(gpu, planning, sc, pl, res, scales, c=11)


def test_cfg5_full_size_64ch_ambience_folded_fx(gpu):
    """BASELINE configs[4] on one GPU: 128 static events x 64 capsules, 4 s RIRs (the planner's choice for a batch of this size:
    12 partitions of 16384 through the quad-tile transforms of csrc/al_quad16.h), [Gain, Invert] + peak normalisation folded
    into the clip spectra on the device, a device-drawn white ambience fused into the mixdown."""
    from audiblelight_amd import ambience as amb, plan as planning, synthetic
    from audiblelight_amd.synthesize import _ambience_on_device

    _CFG3.clear()            # the 6.3 GB cfg3 scene is no longer needed
    sc = synthetic.make_scene("cfg5")
    assert sc.irs.shape == (64, 128, 192000) and len(sc.gain_db) == 128 and sc.duration == 60.0
    pl = planning.plan_batch(sc.specs, 64, sc.ir_len, sc.sr)
    assert pl.log2_block == 14 and pl.n_partitions == 12
    res = gpu.render(pl, sc.sources(), sc.irs)
    scales = check_level_invariant(sc, res)
    gains = np.asarray(gpu.mem.download(res.emitter_gain))[:128].astype(np.float64)
    for e, c in ((0, 0), (5, 63), (77, 20), (127, 41)):
        clip = orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(sc.clips[e], sc.gain_db[e])))
        g_ref = orc.emitter_gains(sc.irs[:, [e], :])[0]
        assert gains[e] == pytest.approx(g_ref, rel=2e-5)
        want = fftconvolve(clip.astype(np.float64), sc.irs[c, e].astype(np.float64))[: sc.specs[e].n_samples] * g_ref
        assert_parity(row_of(res, e, c), want, TOL, what=(e, c))
    assert check_every_event_against_the_oracle(sc, res)["events"] == 128
    a = amb.Ambience(channels=64, duration=sc.duration, alias="full", noise="white", ref_db=-65, sample_rate=sc.sr, rng="device", seed=7)
    n_scene = round(sc.duration * sc.sr)
    pair = _ambience_on_device(gpu, a, (64, n_scene))
    row = check_scene_row(gpu, planning, sc, pl, res, scales, c=37, ambience=[pair])
    # the ambience alone sits at the reference's noise floor: mean|scale_c * noise_c| over the scene = 10^(ref_db / 20)
    noise = np.asarray(gpu.mem.download(pair[0])).reshape(64, -1)[:, :n_scene]
    amb_scales = np.asarray(gpu.mem.download(pair[1]))[:64]
    assert np.mean(np.abs(noise * amb_scales[:, None])) == pytest.approx(10 ** (-65 / 20), rel=1e-4)
    assert np.abs(row).max() > 0


def test_cfg4_full_size_scene_batch(gpu):
    """BASELINE configs[3], one rank's share: 8 full-size scenes (32 capsules, 32 events, 1 s RIR, 30 s) through the pipelined
    BatchDriver from host buffers; two of them against the oracle's scenes (every row), three of them again as ONE merged launch
    sequence (batch.render_merged) against the driver's output."""
    from audiblelight_amd import batch, synthetic

    scenes = [synthetic.make_scene("cfg4", scene_index=i) for i in range(8)]
    assert scenes[0].irs.shape == (32, 32, 48000) and scenes[0].duration == 30.0
    jobs = [batch.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration,
                           sample_rate=sc.sr, name=f"s{i}") for i, sc in enumerate(scenes)]
    got = {}
    rep = batch.BatchDriver(gpu).run(jobs, on_scene=got.__setitem__)
    assert rep.n_scenes == 8 and sorted(got) == [f"s{i}" for i in range(8)]
    for i in range(8):
        assert got[f"s{i}"].shape == (32, 1440000) and got[f"s{i}"].dtype == np.float32 and np.isfinite(got[f"s{i}"]).all()
    for i in (0, 5):
        sc = scenes[i]
        spat = [orc.render_event(sc.clips[e], sc.irs[:, [e], :].astype(np.float64), sc.specs[e].snr, sc.specs[e].ref_db, sr=sc.sr)["spatial"]
                for e in range(len(sc.specs))]
        want = orc.mix_scene(spat, list(zip(sc.starts, sc.ends)), sc.duration, sc.sr, keep_padded=False)["scene"]
        for c in range(32):
            assert_parity(got[f"s{i}"][c], want[c], TOL, what=(i, c))
    merged = batch.render_merged(gpu, jobs[1:4])
    for i, m in zip((1, 2, 3), merged):
        assert rel_rms(m, got[f"s{i}"]) < 1e-6
