"""Run the real kernel SOURCE (audiblelight_amd/csrc/al_kernels.hip) compiled for the host by the
test-only emulation layer in tests/hostemu, and compare with the oracle / reference goldens.

This checks the index arithmetic of the HIP code (Stockham passes, real-FFT packing, overlap-save
block bookkeeping, envelope evaluation, level law, mixdown) without a GPU; the -m gpu tests repeat
the same comparisons on the gfx950 build through the same C ABI.
"""
import numpy as np
import pytest

from audiblelight_amd import _hip, engine, plan as planning
from oracle import synth_oracle as orc
from tests import hostemu
from tests.conftest import assert_parity, rel_rms, set_switch

TOL = 1e-4  # BASELINE.json north_star: outputs within 1e-4 relative RMS of the float64 reference


@pytest.fixture(scope="module")
def emu():
    lib = _hip.Library(hostemu.build())
    return engine.Renderer(lib=lib, memory=hostemu.NumpyMemory())


def test_emu_twiddle(emu):
    tw = emu.mem.download(emu.twiddle(10)).reshape(-1, 2)
    k = np.arange(1024)
    np.testing.assert_allclose(tw[:, 0], np.cos(np.pi * k / 1024), atol=1e-7)
    np.testing.assert_allclose(tw[:, 1], -np.sin(np.pi * k / 1024), atol=1e-7)


@pytest.mark.parametrize("log2_block", [10, 11, 12])
def test_emu_static_event_matches_reference(emu, golden, log2_block):
    a, h = golden["g1_audio"], golden["g1_irs"]
    spec = planning.EventSpec(n_samples=len(a), n_emitters=1, snr=10.0)
    pl = planning.plan_batch([spec], n_capsules=4, ir_len=h.shape[2], sample_rate=8000, log2_block=log2_block)
    res = emu.render(pl, [a], h)
    g = emu.mem.download(res.emitter_gain)[:1]
    np.testing.assert_allclose(g, orc.emitter_gains(h.astype(np.float64)), rtol=1e-5)
    raw_ref = golden["g1_full_conv"][:, : len(a)] * orc.emitter_gains(h.astype(np.float64))[0]
    assert_parity(res.raw_spatial(0), raw_ref, TOL)
    assert_parity(res.spatial_audio(0), golden["g1_spatial"], TOL)
    res.check_finite()


@pytest.mark.parametrize("log2_block", [13, 14])
def test_emu_narrow_transforms_at_large_blocks(emu, golden, monkeypatch, log2_block):
    """B >= 8192 defaults to 32 complex values per thread; AL_FLAG_NARROW_FFT keeps the 16-value kernels reachable."""
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", "4")   # AL_FLAG_NARROW_FFT
    a, h = golden["g1b_audio"], golden["g1b_irs"]
    pl = planning.plan_batch([planning.EventSpec(n_samples=len(a), n_emitters=1, snr=0.5, ref_db=-50)],
                             n_capsules=3, ir_len=h.shape[2], sample_rate=8000, log2_block=log2_block)
    assert_parity(emu.render(pl, [a], h).spatial_audio(0), golden["g1b_spatial"], TOL)


@pytest.mark.parametrize("log2_block", [10, 13])
def test_emu_runs_of_blocks_per_workgroup(emu, golden, monkeypatch, log2_block):
    """AL_FLAG_SYNTH_RUN / AL_FLAG_IR_RUN: several output blocks / IR partitions per workgroup (loop + prefetch)."""
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", str((3 << 16) | (2 << 24)))
    a, h = golden["g1_audio"], golden["g1_irs"]
    pl = planning.plan_batch([planning.EventSpec(n_samples=len(a), n_emitters=1, snr=10.0)], 4, h.shape[2], 8000,
                             log2_block=log2_block)
    assert_parity(emu.render(pl, [a], h).spatial_audio(0), golden["g1_spatial"], TOL)


def test_emu_moving_event_wide_transforms(emu, golden):
    """A moving event (envelope-weighted signal spectra) through the 32-values-per-thread kernels (B = 8192)."""
    a, h = golden["g3b_audio"], golden["g3b_irs"]
    spec = planning.EventSpec(n_samples=len(a), n_emitters=5, snr=12.0, is_moving=True, duration=len(a) / 8000)
    pl = planning.plan_batch([spec], 4, h.shape[2], 8000, log2_block=13)
    assert_parity(emu.render(pl, [a], h).spatial_audio(0), golden["g3b_spatial"], TOL)


@pytest.mark.parametrize("log2_block", [13, 14])
def test_emu_large_blocks(emu, golden, log2_block):
    a, h = golden["g1b_audio"], golden["g1b_irs"]  # clip shorter than the IR
    pl = planning.plan_batch([planning.EventSpec(n_samples=len(a), n_emitters=1, snr=0.5, ref_db=-50)],
                             n_capsules=3, ir_len=h.shape[2], sample_rate=8000, log2_block=log2_block)
    res = emu.render(pl, [a], h)
    assert_parity(res.spatial_audio(0), golden["g1b_spatial"], TOL)


def test_emu_zero_emitter_event(emu, golden):
    a = golden["g2_audio"]
    pl = planning.plan_batch([planning.EventSpec(n_samples=len(a), n_emitters=0, snr=7.0)],
                             n_capsules=4, ir_len=100, sample_rate=8000, log2_block=10)
    res = emu.render(pl, [a], np.zeros((4, 0, 100)))
    assert rel_rms(res.spatial_audio(0), golden["g2_spatial"]) < 1e-6


@pytest.mark.parametrize("tag,n_ir", [("g3a", 3), ("g3b", 5)])
def test_emu_moving_event(emu, golden, tag, n_ir):
    a, h = golden[f"{tag}_audio"], golden[f"{tag}_irs"]
    spec = planning.EventSpec(n_samples=len(a), n_emitters=n_ir, snr=12.0, is_moving=True, duration=len(a) / 8000)
    pl = planning.plan_batch([spec], n_capsules=4, ir_len=h.shape[2], sample_rate=8000, log2_block=10)
    res = emu.render(pl, [a], h)
    raw_ref = orc.fit_length(golden[f"{tag}_raw"], len(a))
    assert_parity(res.raw_spatial(0), raw_ref, TOL)
    assert_parity(res.spatial_audio(0), golden[f"{tag}_spatial"], TOL)


def test_emu_moving_event_sliding_window_kernel(emu, golden):
    """Short streams (n_j <= 4 blocks) take k_spectral_mac_moving; result must match the reference as well."""
    a, h = golden["g3b_audio"], golden["g3b_irs"]
    spec = planning.EventSpec(n_samples=len(a), n_emitters=5, snr=12.0, is_moving=True, duration=len(a) / 8000)
    pl = planning.plan_batch([spec], 4, h.shape[2], 8000, log2_block=12)
    assert pl.events["reserved"][0] == 1 and int(pl.streams["n_j"].max()) <= 4
    assert_parity(emu.render(pl, [a], h).spatial_audio(0), golden["g3b_spatial"], TOL)
    # and next to a static event + a long-stream moving event in the same batch (both kernels run)
    a2, h2 = golden["g1_audio"], golden["g1_irs"]
    a3, h3 = golden["g3a_audio"], golden["g3a_irs"][:, :, :1200]
    specs = [planning.EventSpec(len(a2), 1, 10.0, emitter0=0), spec.__class__(len(a), 5, 12.0, emitter0=1, is_moving=True,
                                                                                 duration=len(a) / 8000)]
    irs = np.concatenate([h2[:, :, :1200], h], axis=1)
    pl2 = planning.plan_batch(specs, 4, 1200, 8000, log2_block=12)
    res = emu.render(pl2, [a2, a], irs)
    assert_parity(res.spatial_audio(1), golden["g3b_spatial"], TOL)
    want = orc.render_event(a2, irs[:, :1, :].astype(np.float64), 10.0, sr=8000)["spatial"]
    assert_parity(res.spatial_audio(0), want, TOL)


def test_emu_full_scene_with_ambience(emu, golden):
    sr, dur, C = 8000, 2.0, 4
    specs, clips, irs, col = [], [], [], 0
    for i, (na, ne, st, snr, mv, dry) in enumerate(golden["g8_specs"]):
        a = golden[f"g8_audio{i}"]
        clips.append(a)
        irs.append(golden[f"g8_irs{i}"])
        specs.append(planning.EventSpec(n_samples=len(a), n_emitters=int(ne), snr=float(snr), emitter0=col,
                                        is_moving=bool(mv), duration=len(a) / sr))
        col += int(ne)
    mic_ir = np.concatenate(irs, axis=1)
    pl = planning.plan_batch(specs, n_capsules=C, ir_len=mic_ir.shape[2], sample_rate=sr, log2_block=10)
    res = emu.render(pl, clips, mic_ir)
    for i in range(len(specs)):
        assert_parity(res.spatial_audio(i), golden[f"g8_spatial{i}"], TOL)
    # chunked execution over a reused workspace gives bit-identical results
    for chunk in (1, 2):
        res_c = emu.render(pl, clips, mic_ir, chunk_events=chunk)
        np.testing.assert_array_equal(res_c.scales(), res.scales())
        for i in range(len(specs)):
            np.testing.assert_array_equal(res_c.raw_spatial(i), res.raw_spatial(i))
    starts = [float(s[2]) for s in golden["g8_specs"]]
    ends = [s + len(c) / sr for s, c in zip(starts, clips)]
    mix = planning.plan_mixdown(starts, ends, [len(c) for c in clips], [C] * 5, pl.events["out_off"],
                                list(range(5)), dur, sr, C)
    amb = golden["g8_ambience"].astype(np.float32)
    amb_dev = emu.mem.upload(amb.reshape(-1))
    stats = emu.mem.download(emu.row_stats(amb_dev, 1, amb.size)).reshape(-1, 4)
    assert stats[0, 0] == pytest.approx(np.abs(amb.astype(np.float64)).sum(), rel=1e-6)
    assert stats[0, 1] == pytest.approx(np.abs(amb).max())
    mult = np.float32(orc.db_gain(-65, stats[0, 0] / amb.size))
    scene = emu.mem.download(emu.mixdown(mix, res, ambience=[(amb_dev, emu.mem.upload(np.full(mix.n_capsules, mult, np.float32)))]))
    scene = scene[: C * mix.n_samples].reshape(C, mix.n_samples)
    assert_parity(scene, golden["g8_scene"], TOL)
    # without ambience: plain overwrite path
    scene2 = emu.mem.download(emu.mixdown(mix, res))[: C * mix.n_samples].reshape(C, mix.n_samples)
    want = golden["g8_scene"].astype(np.float64) - mult * amb.astype(np.float64)
    assert rel_rms(scene2, want) < 2e-4


def test_energy_only_batches_refuse_the_spectra_stages(emu):
    """Renderer.prepare(spectra_workspaces=False) -- the general STFT path's batch: IR energy pass + level law only -- allocates one
    block per spectra workspace; running a transform / accumulate stage on it would write far past them, so the batch refuses."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal(5000).astype(np.float32)
    h = rng.standard_normal((2, 3, 700)).astype(np.float32)
    spec = planning.EventSpec(n_samples=len(a), n_emitters=3, snr=5.0, is_moving=True, duration=len(a) / 8000)
    pl = planning.plan_batch([spec], 2, 700, 8000, log2_block=10)
    batch = emu.prepare(pl, [a], h, emitter_parts=np.zeros(3, dtype=np.int32), spectra_workspaces=False)
    assert batch.energy_only and len(batch.bufs["yspec"]) <= 2 * 2 * 1024 + 16
    batch.run_stage("al_ir_spectra")
    batch.run_stage("al_emitter_gains")
    gains = np.asarray(emu.mem.download(batch.bufs["emitter_gain"]))[:3]
    np.testing.assert_allclose(gains, orc.emitter_gains(h), rtol=2e-5)
    for call in (lambda: batch.run(), lambda: batch.run_stage("al_spectral_mac"), lambda: batch.run(stages=("al_block_synthesis",))):
        with pytest.raises(RuntimeError, match="without spectra workspaces"):
            call()
