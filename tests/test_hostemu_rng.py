"""Device-side normal draws for the ambience (csrc/al_rng.h; Ambience(rng="device")): the reference's OWN acceptance tests of
``powerlaw_psd_gaussian`` (tests/test_ambience.py:30-76: variance, small-sample variance, spectral slope, cumulative scaling,
fixed-seed reproducibility) and of ``Ambience.load_ambience`` (:107-138: shape, unit per-channel peak), ported, plus the
generator's known-answer vectors.  Here on the host-emulated kernels at reduced sizes; tests/test_gpu_rng.py runs the same
scenarios on the MI355X at the reference's sizes."""
import ctypes as ct

import numpy as np
import pytest

from audiblelight_amd import _hip, ambience as amb, core, engine, synthesize as syn
from oracle import synth_oracle as orc
from tests import hostemu
from tests.conftest import rel_rms


@pytest.fixture(scope="module", autouse=True)
def emu_renderer():
    r = engine.Renderer(lib=_hip.Library(hostemu.build()), memory=hostemu.NumpyMemory())
    syn.set_renderer(r)
    yield r
    syn.set_renderer(None)


def on_gpu() -> bool:
    return hasattr(syn.get_renderer().mem, "torch")


def test_philox_known_answers():
    """Philox-4x32-10 against the Random123 known-answer vectors (kat_vectors: philox4x32 10), through the C ABI's host
    entry point, which calls the very function the kernels call."""
    lib = syn.get_renderer().lib
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        out = (ct.c_uint32 * 4)()
        lib.call("al_philox4x32_10", (ct.c_uint32 * 4)(*ctr), (ct.c_uint32 * 2)(*key), out)
        assert tuple(out) == want


def test_normal_fill_is_the_documented_function_of_seed_and_index():
    """al_normal_fill: element i = normal (i % 4) of Philox block i / 4 under (seed, tag), Box-Muller as csrc/al_rng.h states it
    -- restated here in numpy; independent of n (the launch geometry) and reproducible."""
    r = syn.get_renderer()
    n, seed, tag = 1003, 0x123456789ABCDEF, 7
    out = r.mem.empty(n + 1)
    r.lib.call("al_normal_fill", r.mem.ptr(out), n, ct.c_uint64(seed), tag, 1.0, r.mem.stream())
    r.mem.synchronize()
    got = r.mem.download(out)[:n]
    want = np.empty(4 * ((n + 3) // 4))
    for q in range((n + 3) // 4):
        x = orc.philox4x32_10((q & 0xffffffff, q >> 32, tag, 0), (seed & 0xffffffff, seed >> 32))
        for h in range(2):
            u1 = (np.float32(x[2 * h]) + np.float32(0.5)) * np.float32(2.0 ** -32)
            u2 = np.float32(x[2 * h + 1]) * np.float32(2.0 ** -32)
            rad = np.sqrt(-2.0 * np.log(np.float64(u1)))
            want[4 * q + 2 * h] = rad * np.cos(2 * np.pi * np.float64(u2))
            want[4 * q + 2 * h + 1] = rad * np.sin(2 * np.pi * np.float64(u2))
    np.testing.assert_allclose(got, want[:n], rtol=2e-5, atol=2e-6)
    short = r.mem.empty(64)
    r.lib.call("al_normal_fill", r.mem.ptr(short), 37, ct.c_uint64(seed), tag, 1.0, r.mem.stream())
    r.mem.synchronize()
    np.testing.assert_array_equal(r.mem.download(short)[:37], got[:37])


@pytest.mark.parametrize("exponent", [0, 0.5, 1, 2])
def test_var_distribution(exponent):       # reference tests/test_ambience.py:30-37 (+ the white shortcut, exponent 0)
    size = (100, 2 ** 16) if on_gpu() else (24, 2 ** 11)
    y = amb.powerlaw_psd_gaussian(exponent, size, fmin=0, seed=1, rng="device")
    ystd = y.std(axis=-1)
    assert (abs(1 - ystd) < 3 * ystd.std()).mean() > 0.95
    if exponent == 0:
        assert abs(ystd.mean() - 1) < 0.05 and abs(y.mean()) < 0.05


@pytest.mark.parametrize("nsamples", [10, 11])
def test_small_sample_var(nsamples):       # :40-43 (both parities of n: the odd one takes the Hermitian-extension route)
    shape = (500, 500, nsamples) if on_gpu() else (40, 20, nsamples)
    ystd = amb.powerlaw_psd_gaussian(0.5, shape, seed=1, rng="device").std(axis=-1)
    assert (abs(1 - ystd) < 3 * ystd.std()).mean() > 0.95
    white = amb.powerlaw_psd_gaussian(0, shape, seed=1, rng="device").std(axis=-1)
    assert (abs(1 - white) < 3 * white.std()).mean() > 0.95


@pytest.mark.parametrize("exponent", [0, 0.5, 1, 2])
def test_slope_distribution(exponent):     # :46-58
    size = (100, 2 ** 16) if on_gpu() else (24, 2 ** 11)
    y = amb.powerlaw_psd_gaussian(exponent, size, fmin=0, seed=1, rng="device")
    yfft = np.fft.fft(y)
    f = np.fft.fftfreq(y.shape[-1])
    m = f > 0
    fit, fcov = np.polyfit(np.log10(f[m]), np.log10(np.abs(yfft[..., m].T ** 2)), 1, cov=True)
    assert (exponent + fit[0] < 3 * np.sqrt(fcov[0, 0])).mean() > 0.95
    assert abs(np.mean(fit[0]) + exponent) < 0.1        # two-sided: the mean fitted slope IS -exponent


def test_cumulative_scaling():             # :61-67
    n_repeats, n_steps = (1000, 100) if on_gpu() else (300, 100)
    for beta in (0, 1e-9):                 # the time-domain shortcut and the irfft route (a beta that is white to 1e-9)
        y = amb.powerlaw_psd_gaussian(beta, (n_repeats, n_steps), seed=1, rng="device")
        msd = (y.sum(axis=-1) ** 2).mean(axis=0)
        se = (y.sum(axis=-1) ** 2).std(axis=0) / np.sqrt(n_repeats)
        assert abs(n_steps - msd) < 3 * se


def test_random_state_reproducibility():   # :70-76
    for exp in (0, 1):
        y1 = amb.powerlaw_psd_gaussian(exp, 5, seed=1, rng="device")
        np.random.seed(123)
        y2 = amb.powerlaw_psd_gaussian(exp, 5, seed=1, rng="device")
        np.testing.assert_array_equal(y1, y2)
        assert not np.array_equal(y1, amb.powerlaw_psd_gaussian(exp, 5, seed=2, rng="device"))
    a = amb.powerlaw_psd_gaussian(1, (3, 64), seed=9, rng="device")
    assert not np.allclose(a[0], a[1])     # rows are independent streams


def test_normality_of_the_draws():
    """Moments and tail mass of the Box-Muller normals (the reference relies on numpy's ziggurat for these)."""
    n = 4_000_000 if on_gpu() else 200_000
    x = amb.powerlaw_psd_gaussian(0, n, seed=3, rng="device") * amb._flat_sigma(n) / np.sqrt(2.0 / n)   # undo the 1/sigma scale: raw N(0,1)
    assert abs(x.mean()) < 4 / np.sqrt(n) and abs(x.var() - 1) < 4 * np.sqrt(2.0 / n)
    assert abs(np.mean(x ** 3)) < 4 * np.sqrt(15.0 / n) and abs(np.mean(x ** 4) - 3) < 4 * np.sqrt(96.0 / n)
    for k, p in ((1, 0.31731), (2, 0.045500), (3, 0.0026998)):
        frac = np.mean(np.abs(x) > k)
        assert abs(frac - p) < 5 * np.sqrt(p * (1 - p) / n)
    lag = np.mean(x[:-1] * x[1:])
    assert abs(lag) < 4 / np.sqrt(n)


@pytest.mark.parametrize("noise", ["gaussian", "white", 2.0])
@pytest.mark.parametrize("normalize", [True, False])
def test_ambience_cls(noise, normalize):   # :107-138
    cls = amb.Ambience(4, 0.25, noise=noise, alias="tester", sample_rate=8000, rng="device")
    assert isinstance(cls.to_dict(), dict) and cls.to_dict().get("rng") == "device"
    before = cls.to_dict()
    twin = amb.Ambience(4, 0.25, noise=noise, alias="tester", sample_rate=8000, rng="device")
    assert cls == twin          # equal at construction (the reference compares dictionaries; device_seed is not part of equality)
    loaded = cls.load_ambience(normalize=normalize)
    # to_dict is pure: the same before and after audio exists (a "gaussian" object holds its key from construction on) ...
    assert cls.to_dict() == before and cls.to_dict()["noise_kwargs"] == {} and cls == twin
    assert ("device_seed" in before) == (noise == "gaussian")
    assert loaded.shape == (4, 2000)
    for channel in loaded:
        if normalize:
            assert pytest.approx(np.max(np.abs(channel))) == 1.0
        else:
            assert not pytest.approx(np.max(np.abs(channel))) == 1.0
    again = amb.Ambience.from_dict(cls.to_dict())
    assert again.rng == "device"
    np.testing.assert_array_equal(again.load_ambience(normalize=normalize), loaded)


def test_scene_with_device_drawn_ambience_matches_the_oracle_given_the_same_noise():
    """Scene.generate with a device-drawn ambience: the noise never meets the host (un-normalised buffer + per-channel
    multipliers from al_ambience_scales, added inside the mixdown); given THAT noise the scene equals the oracle's mix
    (ambience.py:211-214 peak normalisation, synthesize.py:350-356 floor multiplier)."""
    rng = np.random.default_rng(5)
    sr, C, L = 8000, 3, 300
    irs = (rng.standard_normal((C, 2, L)) * np.exp(-np.arange(L) / 60.0)).astype(np.float32)
    scene = core.Scene(1.0, core.StaticIRState({"mic000": irs}), sample_rate=sr, ref_db=-60)
    clips = [rng.standard_normal(n).astype(np.float32) for n in (3000, 2500)]
    for i, c in enumerate(clips):
        scene.add_event(core.Event(f"e{i}", c, sr, snr=10.0 + i, scene_start=0.1 + 0.2 * i))
    a = amb.Ambience(C, 1.0, alias="a", noise="pink", ref_db=-55, sample_rate=sr, rng="device", seed=11)
    scene.add_ambience(a)
    got = scene.generate()["mic000"]
    assert a.audio is None                                   # nobody downloaded the noise
    noise = amb.Ambience(C, 1.0, alias="a", noise="pink", ref_db=-55, sample_rate=sr, rng="device", seed=11).load_ambience()
    spat = [orc.render_event(orc.peak_normalise_clip(c), irs[:, [i], :].astype(np.float64), 10.0 + i, ref_db=-60, sr=sr)["spatial"]
            for i, c in enumerate(clips)]
    want = orc.mix_scene(spat, [(e.scene_start, e.scene_end) for e in scene.events.values()], 1.0, sr,
                         ambiences=[(noise, -55)], keep_padded=False)["scene"]
    assert rel_rms(got, want) < 1e-4
