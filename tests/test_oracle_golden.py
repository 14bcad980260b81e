"""Pin the CPU oracle (oracle/synth_oracle.py) to the real reference's outputs.

The golden vectors were produced by importing /root/reference (tests/golden/make_golden.py).
Tolerance: the oracle is float64 like the reference, so 1e-10 relative.
"""
import os

import numpy as np
import pytest

from oracle import synth_oracle as orc
from tests.conftest import rel_rms

TOL = 1e-10


def test_g9_apply_snr_and_db_known_answers(golden):
    # reference tests/test_synthesize.py:42-57
    expected_max = [2.0, 1.0, 6.0, 0.5, 0.0]
    for i, snr in enumerate(golden["g9_snr_vals"]):
        got = orc.snr_scale(golden[f"g9_snr_in{i}"], float(snr))
        np.testing.assert_allclose(got, golden[f"g9_snr_out{i}"], rtol=1e-15, atol=0)
        assert np.isclose(np.max(np.abs(got)), expected_max[i], rtol=1e-5)
    # reference tests/test_synthesize.py:307-337
    expected = [1.0, 2.0, 0.5, 100.0, 0.01]
    for d, x, m, e in zip(golden["g9_db"], golden["g9_level"], golden["g9_mult"], expected):
        got = orc.db_gain(float(d), float(x))
        assert got == pytest.approx(float(m), rel=1e-15)
        assert np.isclose(got, e, atol=1e-4)


def test_g5_normalize_irs(golden):
    irs = golden["g5_irs"].astype(np.float64)
    got = orc.unit_energy_irs(irs.transpose(1, 0, 2)).transpose(1, 0, 2)
    assert rel_rms(got, golden["g5_norm"]) < TOL
    # invariant from reference tests/test_synthesize.py:365-370
    assert np.mean(np.sqrt((got ** 2).sum(-1))) == pytest.approx(1.0)
    g = orc.emitter_gains(irs)
    assert rel_rms(irs * g[None, :, None], golden["g5_norm"]) < TOL


def test_g1_static(golden):
    a, h = golden["g1_audio"], golden["g1_irs"].astype(np.float64)
    full = orc.convolve_static(a.astype(np.float64), h[:, 0].T)
    assert full.shape == golden["g1_full_conv"].shape
    assert rel_rms(full, golden["g1_full_conv"]) < TOL
    res = orc.render_event(a, h, snr=10.0, ref_db=-65, sr=8000)
    assert rel_rms(res["spatial"], golden["g1_spatial"]) < TOL
    # composite level law invariant
    assert np.mean(np.abs(res["spatial"])) == pytest.approx(10 ** ((-65 + 10.0) / 20), rel=1e-12)


def test_g1b_static_ir_longer_than_clip(golden):
    res = orc.render_event(golden["g1b_audio"], golden["g1b_irs"].astype(np.float64), snr=0.5, ref_db=-50, sr=8000)
    assert res["spatial"].shape == golden["g1b_spatial"].shape
    assert rel_rms(res["spatial"], golden["g1b_spatial"]) < TOL


def test_g2_zero_emitters(golden):
    res = orc.render_event(golden["g2_audio"], np.zeros((4, 0, 100)), snr=7.0, ref_db=-65, sr=8000)
    assert rel_rms(res["spatial"], golden["g2_spatial"]) < TOL


@pytest.mark.parametrize("tag,n_ir", [("g3a", 3), ("g3b", 5)])
def test_g3_moving_both_forms(golden, tag, n_ir):
    a, h = golden[f"{tag}_audio"], golden[f"{tag}_irs"].astype(np.float64)
    dur = len(a) / 8000
    w, n_frames = orc.tv_frames(len(a), dur, n_ir, 8000)
    np.testing.assert_allclose(w, golden[f"{tag}_w"], atol=1e-15)
    hn = orc.unit_energy_irs(h.transpose(1, 0, 2)).transpose(1, 0, 2)
    raw_ref = golden[f"{tag}_raw"]
    raw_stft = orc.convolve_moving_stft(a, hn, dur, 8000)
    raw_env = orc.convolve_moving(a, hn, dur, 8000)
    assert raw_stft.shape == raw_ref.shape == raw_env.shape
    assert rel_rms(raw_stft, raw_ref) < TOL
    assert rel_rms(raw_env, raw_ref) < TOL  # the envelope identity the HIP path relies on
    for impl in ("envelope", "stft"):
        res = orc.render_event(a, h, snr=12.0, ref_db=-65, is_moving=True, duration=dur, sr=8000, moving_impl=impl)
        assert rel_rms(res["spatial"], golden[f"{tag}_spatial"]) < TOL


def test_g4_dry_path(golden):
    res = orc.render_event(golden["g4_audio"], golden["g4_irs"].astype(np.float64), snr=9.0, ref_db=-65, sr=8000,
                           ref_ir_channel=0, direct_path_time_ms=[5, 60])
    assert rel_rms(res["spatial"], golden["g4_spatial"]) < TOL
    assert res["dry"].shape == golden["g4_dry"].shape
    assert rel_rms(res["dry"], golden["g4_dry"]) < TOL


@pytest.mark.parametrize("beta", [0, 1, 2, -1])
@pytest.mark.parametrize("n", [1000, 1001])
def test_g6_powerlaw(golden, beta, n):
    got = orc.powerlaw_noise(beta, (4, n))
    np.testing.assert_allclose(got, golden[f"g6_b{beta}_n{n}"], rtol=1e-12, atol=1e-13)


def test_g6_powerlaw_fmin_seed_and_1d(golden):
    np.testing.assert_allclose(orc.powerlaw_noise(1, (2, 512), fmin=0.1, seed=7), golden["g6_fmin"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(orc.powerlaw_noise(1, 300), golden["g6_1d"], rtol=1e-12, atol=1e-13)
    with pytest.raises(ValueError):
        orc.powerlaw_noise(1, 16, fmin=0.7)


def test_g7_ambience(golden):
    noise = orc.ambience_noise(1, 4, 0.5, 8000)
    np.testing.assert_allclose(noise, golden["g7_noise"], rtol=1e-12, atol=1e-13)
    # per-channel peak == 1 (reference tests/test_ambience.py:134-136)
    np.testing.assert_allclose(np.abs(noise).max(axis=1), 1.0, rtol=1e-12)
    assert orc.db_gain(-60, np.mean(np.abs(noise))) == pytest.approx(float(golden["g7_mult"]), rel=1e-13)


def test_g8_full_scene(golden):
    specs = golden["g8_specs"]
    sr, dur = 8000, 2.0
    spatials, slots, dries = [], [], []
    for i, (na, ne, st, snr, mv, dry) in enumerate(specs):
        a = golden[f"g8_audio{i}"]
        h = golden[f"g8_irs{i}"].astype(np.float64)
        res = orc.render_event(a, h, snr=float(snr), ref_db=-65, is_moving=bool(mv), duration=len(a) / sr, sr=sr,
                               ref_ir_channel=1 if dry else None, direct_path_time_ms=[2, 20] if dry else None)
        assert rel_rms(res["spatial"], golden[f"g8_spatial{i}"]) < TOL
        spatials.append(res["spatial"])
        slots.append((float(st), float(st) + len(a) / sr))
        dries.append(res["dry"])
    amb = orc.ambience_noise(0, 4, dur, sr)
    np.testing.assert_allclose(amb, golden["g8_ambience"], rtol=1e-12, atol=1e-13)
    mix = orc.mix_scene(spatials, slots, dur, sr, ambiences=[(amb, -65)], dries=dries)
    assert mix["scene"].dtype == np.float32 and mix["scene"].shape == golden["g8_scene"].shape
    assert rel_rms(mix["scene"], golden["g8_scene"]) < 1e-6  # float32 accumulation buffer
    for i in range(len(specs)):
        assert rel_rms(mix["padded"][i], golden[f"g8_padded{i}"]) < 1e-6
    assert rel_rms(dries[4], golden["g8_dry4"]) < TOL
    assert rel_rms(mix["dry_padded"][4], golden["g8_dry_padded4"]) < 1e-6


def test_fade_endpoints_invert_reverse():
    # reference tests/test_augmentation.py:300-327, 504-515, 518-532
    for shape in ("linear", "exponential", "logarithmic", "quarter_sine", "half_sine"):
        y = orc.fx_fade(np.ones(8000), 8000, 0.25, 0.25, shape, shape)
        assert abs(y[0]) < 1e-6 and abs(y[-1]) < 1e-6 and y[4000] == pytest.approx(1.0)
    assert np.array_equal(orc.fx_invert(np.ones(10)), -np.ones(10))
    x = np.arange(10.0)
    assert orc.fx_reverse(x)[0] == 9 and orc.fx_reverse(x)[-1] == 0


def test_c_direct_form_witness(golden):
    """oracle/conv_direct.c (time-domain, no FFT, plain C) against the reference goldens: the full convolution of G1,
    the truncated one of G1b (IR longer than the clip), and the float32 in-place mixdown adds of G8."""
    from oracle import conv_direct as cd

    a, h = golden["g1_audio"], golden["g1_irs"].astype(np.float64)
    full = cd.conv_direct(a, h[:, 0], len(a) + h.shape[2] - 1)
    assert rel_rms(full, golden["g1_full_conv"]) < 1e-12   # that golden was made from a float64 clip
    a, h = golden["g1b_audio"], golden["g1b_irs"].astype(np.float64)
    raw = cd.conv_direct(a, h[:, 0] * orc.emitter_gains(h)[0], len(a))
    scaled, _ = orc.level_law(raw, 0.5, -50)
    # the reference hands scipy a float32 clip, whose spectrum scipy then takes in single precision: its own output
    # sits ~1e-7 from the exact (time-domain, float64) sum
    assert rel_rms(scaled, golden["g1b_spatial"]) < 5e-7
    # mixdown: ambience first, then the five events in insertion order, one float32 rounding per add
    sr, n_scene = 8000, golden["g8_scene"].shape[1]
    amb = golden["g8_ambience"]
    scene = np.zeros((4, n_scene), dtype=np.float32)
    scene += (orc.db_gain(-65, np.mean(np.abs(amb))) * amb).astype(np.float64)
    for i, spec in enumerate(golden["g8_specs"]):
        x = golden[f"g8_spatial{i}"]
        lo, hi = orc.event_slot(float(spec[2]), float(spec[2]) + x.shape[1] / sr, sr, n_scene)
        cd.mix_add(scene, x, lo, min(hi - lo, x.shape[1]))
    np.testing.assert_allclose(scene, golden["g8_scene"], rtol=0, atol=1e-7 * np.abs(golden["g8_scene"]).max())


def test_g10_stft_intermediates():
    """Oracle restatement of stft / perform_time_variant_convolution / istft_overlap_synthesis (synthesize.py:109-274)
    against the reference's own outputs (tests/golden/reference_stft_vectors.npz, complex64 storage)."""
    import os

    from scipy import fft as sp_fft

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_stft_vectors.npz"))
    a, h = z["g10_audio"].astype(np.float64), z["g10_irs"].astype(np.float64)
    assert np.abs(orc.stft_frames(a) - z["g10_stft_audio"]).max() < 1e-5 * np.abs(z["g10_stft_audio"]).max()
    assert np.abs(orc.stft_frames(h) - z["g10_stft_irs"]).max() < 1e-5 * np.abs(z["g10_stft_irs"]).max()
    assert np.abs(orc.stft_frames(a[:777], 256, 128, 64) - z["g10_stft_b"]).max() < 1e-5 * np.abs(z["g10_stft_b"]).max()
    # full chain: STFT-domain convolution restated literally == reference istft output
    got = orc.convolve_moving_stft(a, h, len(a) / 8000, 8000)
    assert rel_rms(got.T, z["g10_istft"]) < 1e-9


def _fx_cases():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_fx_vectors.npz"))
    return z, int(z["sr"])


def test_g11_reference_fx_vectors_pin_the_oracle():
    """G11 (tests/golden/make_fx_golden.py): Fade (every shape, fades longer than the clip), Invert, Reverse and the four
    TimeWarp classes as the reference's OWN numpy code computes them (augmentation.py:1403-1790), through
    Augmentation.process' wrap contract.  The oracle's restatements must reproduce them exactly (float64 arithmetic on
    the same float32 input)."""
    import random

    z, sr = _fx_cases()
    for i, case in enumerate(z["fade_cases"]):
        a, b, la, lb, src = str(case).split(",")
        x = z["x_short" if src == "short" else "x"]
        want = z[f"fade_{i}"]
        got = orc.fx_wrap(lambda v: orc.fx_fade(v, sr, float(la), float(lb), a, b), x)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-15, err_msg=str(case))
    np.testing.assert_array_equal(orc.fx_wrap(orc.fx_invert, z["x"]), z["invert"])
    np.testing.assert_array_equal(orc.fx_wrap(orc.fx_reverse, z["x"]), z["reverse"])
    modes = {"TimeWarpSilence": "silence", "TimeWarpDuplicate": "duplicate", "TimeWarpRemove": "remove", "TimeWarpReverse": "reverse"}
    for i, case in enumerate(z["tw_cases"]):
        name, fps, prob, src, seed = str(case).split(",")
        x = z[src]
        fl = round(sr / float(fps))
        rows = 1 if fl > len(x) else fl
        random.seed(int(seed))
        decisions = [random.random() < float(prob) for _ in range(rows)]
        got = orc.fx_wrap(lambda v: orc.fx_timewarp(v, sr, float(fps), decisions, modes[name]), x)
        np.testing.assert_array_equal(got, z[f"tw_{i}"], err_msg=str(case))


GEOMETRY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_geometry_vectors.npz")


def test_g14_other_stft_geometries():
    """G14 (tests/golden/make_golden.py::geometry_vectors): the reference's render of one moving event under seven STFT
    geometries -- win != 2*hop, fft < 2*win - 1, 75 % overlap, a scaled default, win > fft/2 with hop = win/3, non-power-of-two
    sizes, fft < win -- reproduced by the oracle's literal STFT-domain restatement to 1e-10; where the envelope identity holds
    (win == 2*hop, fft >= 2*win - 1: the scaled default) the envelope form agrees as well."""
    with np.load(GEOMETRY) as z:
        z = {k: z[k] for k in z.files}
    a, h = z["g14_audio"], z["g14_irs"].astype(np.float64)
    assert len(z["g14_geometries"]) == 7 and len(z["g14_refused"]) == 3
    for fft_size, win, hop in z["g14_geometries"].tolist():
        tag = f"g14_{fft_size}_{win}_{hop}"
        got = orc.render_event(a, h, 11.0, ref_db=-65, is_moving=True, duration=len(a) / 8000, sr=8000, nfft=fft_size, win=win,
                               hop=hop)["spatial"]
        assert rel_rms(got, z[tag + "_spatial"]) < 1e-10, tag
        raw = orc.convolve_moving_stft(a, orc.unit_energy_irs(h.transpose(1, 0, 2)).transpose(1, 0, 2), len(a) / 8000, 8000,
                                       fft_size, win, hop)
        assert raw.shape == z[tag + "_raw"].shape and rel_rms(raw, z[tag + "_raw"]) < 1e-10, tag
        if win == 2 * hop and fft_size >= 2 * win - 1:
            env = orc.render_event(a, h, 11.0, ref_db=-65, is_moving=True, duration=len(a) / 8000, sr=8000, nfft=fft_size,
                                   win=win, hop=hop, moving_impl="envelope")["spatial"]
            assert rel_rms(env, z[tag + "_spatial"]) < 1e-10
    # what the reference refuses: frames longer than the overlap-add buffer allows (fft > 2*hop + win) and win < hop
    for fft_size, win, hop in z["g14_refused"].tolist():
        assert fft_size > 2 * hop + win or win < hop


def test_g15_degenerate_events():
    """G15 (tests/golden/make_golden.py::edge_vectors): snr = 0, a negative snr, an all-zero IR, an all-zero clip, a moving event with
    one all-zero IR, the dry render of an all-zero clip -- the reference keeps silence silent (zeros times 1 / tiny stay zeros in
    float64) and so does the oracle."""
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_edge_vectors.npz")) as z:
        z = {k: z[k] for k in z.files}
    a, h, h3 = z["g15_audio"], z["g15_irs"].astype(np.float64), z["g15_irs_moving"].astype(np.float64)
    cases = {"snr0": (a, h, 0.0, False, {}), "snr_neg": (a, h, -4.0, False, {}), "zero_ir": (a, np.zeros_like(h), 9.0, False, {}),
             "zero_clip": (np.zeros_like(a), h, 9.0, False, {}), "moving_one_zero_ir": (a, h3, 7.0, True, {}),
             "zero_clip_dry": (np.zeros_like(a), h, 9.0, False, dict(ref_ir_channel=0, direct_path_time_ms=[2, 20]))}
    for tag, (clip, irs, snr, moving, kw) in cases.items():
        with np.errstate(all="ignore"):
            res = orc.render_event(clip, irs, snr, ref_db=-65, is_moving=moving, duration=len(clip) / 8000, sr=8000, **kw)
        want = z[f"g15_{tag}_spatial"]
        assert np.isfinite(res["spatial"]).all() and res["spatial"].shape == want.shape
        if np.abs(want).max() == 0:
            assert np.abs(res["spatial"]).max() == 0, tag
        else:
            assert rel_rms(res["spatial"], want) < 1e-10, tag
        if kw:
            assert np.isfinite(res["dry"]).all() and np.abs(res["dry"]).max() == 0 and np.abs(z[f"g15_{tag}_dry"]).max() == 0
