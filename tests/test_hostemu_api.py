"""The reference-named Python API (audiblelight_amd.synthesize / augmentation / core) driven through the
host-emulated kernels: covers the host logic (caching, lazy dicts, errors, mixdown planning) on CPU.
The same scenarios run on the real GPU in tests/test_gpu_api.py."""
import random

import numpy as np
import pytest

from audiblelight_amd import _hip, augmentation as aug, core, engine, synthesize as syn
from oracle import synth_oracle as orc
from tests import hostemu
from tests.conftest import assert_parity, pcm16, rel_rms, set_switch

TOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def emu_renderer():
    r = engine.Renderer(lib=_hip.Library(hostemu.build()), memory=hostemu.NumpyMemory())
    syn.set_renderer(r)
    yield r
    syn.set_renderer(None)


def build_g8_scene(golden, with_ambience=True):
    sr = 8000
    irs, events = [], []
    for i, (na, ne, st, snr, mv, dry) in enumerate(golden["g8_specs"]):
        irs.append(golden[f"g8_irs{i}"])
        events.append(core.Event(f"ev{i}", golden[f"g8_audio{i}"], sr, snr=float(snr), scene_start=float(st),
                                 n_emitters=int(ne), ref_ir_channel=1 if dry else None,
                                 direct_path_time_ms=[2, 20] if dry else None))
    scene = core.Scene(2.0, core.StaticIRState({"mic000": np.concatenate(irs, axis=1)}), sample_rate=sr, ref_db=-65)
    for ev in events:
        scene.add_event(ev)
    if with_ambience:
        class GoldenAmbience:
            alias, ref_db = "a0", -65

            def load_ambience(self, ignore_cache=False, normalize=True):
                return golden["g8_ambience"]
        scene.add_ambience(GoldenAmbience())
    return scene


def test_scene_generate_matches_reference(golden):
    scene = build_g8_scene(golden)
    out = scene.generate()
    assert out["mic000"].dtype == np.float32 and out["mic000"].shape == golden["g8_scene"].shape
    assert_parity(out["mic000"], golden["g8_scene"], TOL)
    for i, ev in enumerate(scene.events.values()):
        assert ev.spatial_audio.is_resident("mic000")           # still in "HBM" until somebody reads it
        got = ev.spatial_audio["mic000"]
        assert isinstance(got, np.ndarray) and not ev.spatial_audio.is_resident("mic000")
        assert_parity(got, golden[f"g8_spatial{i}"], TOL)
        assert_parity(ev._spatial_audio_padded["mic000"], golden[f"g8_padded{i}"], TOL)
    ev4 = scene.events["ev4"]
    assert_parity(ev4._spatial_audio_dry["mic000"], golden["g8_dry4"], TOL)
    assert_parity(ev4._spatial_audio_dry_padded["mic000"], golden["g8_dry_padded4"], TOL)


def test_render_cache_and_host_arrays(golden):
    scene = build_g8_scene(golden, with_ambience=False)
    syn.render_audio_for_all_scene_events(scene)
    ev0 = scene.events["ev0"]
    first = ev0.spatial_audio.device_source("mic000")[0]
    syn.render_audio_for_all_scene_events(scene)                 # cached: nothing re-rendered (synthesize.py:541-542)
    assert ev0.spatial_audio.device_source("mic000")[0] is first
    syn.render_audio_for_all_scene_events(scene, ignore_cache=True)
    assert ev0.spatial_audio.device_source("mic000")[0] is not first
    assert first.keep == ()                                       # a result does not pin the spectra workspace
    # events rendered elsewhere (plain ndarrays in spatial_audio) are mixed as well
    for ev in scene.events.values():
        host = ev.spatial_audio["mic000"]
        ev.spatial_audio = {"mic000": host}
    syn.generate_scene_audio_from_events(scene)
    want = golden["g8_scene"].astype(np.float64) - orc.db_gain(-65, np.mean(np.abs(golden["g8_ambience"]))) * golden["g8_ambience"]
    assert rel_rms(scene.audio["mic000"], want) < 2e-4
    # the mixdown must follow what event.spatial_audio[mic] holds NOW (reference synthesize.py:372-378), not a stale
    # device render: edit one event after rendering, then clear another
    scene2 = build_g8_scene(golden, with_ambience=False)
    syn.render_audio_for_all_scene_events(scene2)
    e1 = scene2.events["ev1"]
    e1.spatial_audio["mic000"] = np.zeros_like(e1.spatial_audio["mic000"])      # user post-processing: silence it
    assert e1.spatial_audio.device_source("mic000") is None
    syn.generate_scene_audio_from_events(scene2)
    sr, n1 = 8000, golden["g8_spatial1"].shape[1]
    lo = round(float(golden["g8_specs"][1][2]) * sr)
    want2 = want.copy()
    want2[:, lo: lo + n1] -= golden["g8_spatial1"]
    assert rel_rms(scene2.audio["mic000"], want2) < 2e-4
    scene2.events["ev0"].clear_audio()
    assert scene2.events["ev0"].spatial_audio.device_source("mic000") is None


def test_render_event_audio_and_errors(golden):
    a, h = golden["g1_audio"], golden["g1_irs"]
    ev = core.Event("g1", a, 8000, snr=10.0)
    syn.render_event_audio(ev, h, "mic000", ref_db=-65)
    assert_parity(ev.spatial_audio["mic000"], golden["g1_spatial"], TOL)
    with pytest.raises(ValueError, match="Moving Event has only one emitter!"):
        syn.render_event_audio(core.Event("m", a, 8000, n_emitters=1, is_moving=True), h, "mic000")
    with pytest.raises(ValueError, match="Expected a moving event!"):
        syn.render_event_audio(core.Event("s", a, 8000, n_emitters=3, is_moving=False), np.repeat(h, 3, 1), "mic000")
    # reference tests/test_synthesize.py:25-39
    with pytest.raises(ValueError, match="Only mono input is supported"):
        syn.time_invariant_convolution(np.ones((5, 2)), np.ones((5, 4)))
    with pytest.raises(ValueError, match="Expected shape of IR should be"):
        syn.time_invariant_convolution(np.ones(5), np.ones(5))
    bad = core.Event("nan", np.array([0.0, np.nan, 1.0], dtype=np.float32), 8000)
    with pytest.raises(ValueError, match="not finite"):
        syn.render_event_audio(bad, h, "mic000")


def test_outputs_carry_the_reference_dtype(golden):
    """A drop-in hands back the reference's dtypes, not only its values (synthesize.py:560-599 under NumPy 2's weak scalars, measured on
    the reference itself: tests/golden/differential_fuzz.py compares dtypes on every array): a clip tiled over the capsules keeps the
    clip's dtype -- float32 from Event.load_audio, the golden G2 is float32 --, a static render and its dry render have fftconvolve's
    result type of clip and IRs, a moving render is float64 whatever went in; Fade and a TimeWarpSilence that silenced a frame
    widen a float32 clip to float64, the other numpy FX keep it."""
    import types

    def duck(audio, n_emitters, **kw):
        return types.SimpleNamespace(alias="d", snr=7.0, sample_rate=8000, is_moving=n_emitters > 1, spatial_audio={}, _spatial_audio_dry={},
                                     duration=len(audio) / 8000,
                                     load_audio=lambda ignore_cache=False, normalize=True: audio, __len__=lambda: n_emitters,
                                     ref_ir_channel=kw.get("ref"), direct_path_time_ms=kw.get("win"))

    a32, h64 = golden["g2_audio"], golden["g1_irs"].astype(np.float64)[:, :, :400]
    assert a32.dtype == np.float32 and golden["g2_spatial"].dtype == np.float32
    cases = [(a32, np.zeros((4, 0, 100)), np.float32), (a32.astype(np.float64), np.zeros((4, 0, 100)), np.float64),
             (a32, h64, np.float64), (a32, h64.astype(np.float32), np.float32), (a32.astype(np.float64), h64.astype(np.float32), np.float64),
             (a32, np.repeat(h64.astype(np.float32), 3, 1), np.float64)]
    for audio, irs, want in cases:
        ev = duck(audio, irs.shape[1], ref=1 if irs.shape[1] == 1 else None, win=[2, 20] if irs.shape[1] == 1 else None)
        syn.render_event_audio(ev, irs, "m", ref_db=-65)
        assert ev.spatial_audio["m"].dtype == want, (audio.dtype, irs.dtype, irs.shape[1])
        if irs.shape[1] == 1:
            assert ev._spatial_audio_dry["m"].dtype == want
    ev = duck(a32, 0)
    syn.render_event_audio(ev, np.zeros((4, 0, 100)), "m", ref_db=-65)
    assert_parity(ev.spatial_audio["m"], golden["g2_spatial"], 1e-6)
    assert aug.Fade(8000, fade_in_len=0.1, fade_out_len=0.1, fade_in_shape="linear", fade_out_shape="linear")(a32).dtype == np.float64
    assert aug.Invert(8000)(a32).dtype == np.float32 and aug.Reverse(8000)(a32).dtype == np.float32
    assert aug.TimeWarpSilence(8000, fps=8.0, prob=1.0)(a32).dtype == np.float64
    assert aug.TimeWarpSilence(8000, fps=8.0, prob=0.0)(a32).dtype == np.float32
    assert aug.TimeWarpReverse(8000, fps=8.0, prob=1.0)(a32).dtype == np.float32


def test_validate_scene_messages(golden):
    scene = build_g8_scene(golden, with_ambience=False)
    syn.validate_scene(scene)
    empty = core.Scene(1.0, core.StaticIRState({"mic000": np.zeros((4, 2, 10))}))
    with pytest.raises(ValueError, match="Scene has no events!"):
        syn.validate_scene(empty)
    with pytest.raises(ValueError, match="WorldState has no emitters!"):
        syn.validate_scene(core.Scene(1.0, core.StaticIRState({"mic000": np.zeros((4, 0, 10))})))


def test_standalone_convolutions(golden):
    a, h = golden["g1_audio"], golden["g1_irs"].astype(np.float64)
    full = syn.time_invariant_convolution(a, h[:, 0].T)
    assert full.shape == golden["g1_full_conv"].shape and rel_rms(full, golden["g1_full_conv"]) < TOL
    a3, h3 = golden["g3a_audio"], golden["g3a_irs"].astype(np.float64)
    hn = orc.unit_energy_irs(h3.transpose(1, 0, 2)).transpose(1, 0, 2)
    ev = core.Event("mv", a3, 8000, n_emitters=3)
    raw = syn.time_variant_convolution(hn, ev)
    assert raw.shape == golden["g3a_raw"].shape and rel_rms(raw, golden["g3a_raw"]) < TOL
    got = syn.normalize_irs(golden["g5_irs"].transpose(1, 0, 2)).transpose(1, 0, 2)
    assert rel_rms(got, golden["g5_norm"]) < 1e-6
    # scalar known answers (reference tests/test_synthesize.py:42-57,307-337)
    assert np.max(np.abs(syn.apply_snr(np.array([0.0, 0.5, -0.5, 1.0, -1.0]), 2.0))) == pytest.approx(2.0)
    assert syn.db_to_multiplier(20.0, 0.1) == pytest.approx(100.0, abs=1e-4)


@pytest.mark.parametrize("n", [8000, 4001])
def test_pointwise_fx_match_definitions(n):
    rng = np.random.default_rng(n)
    x = (0.8 * rng.standard_normal(n)).astype(np.float32)
    sr = 8000
    np.testing.assert_allclose(aug.Gain(sr, gain_db=-4.5)(x), orc.fx_gain(x, -4.5), rtol=1e-6)
    np.testing.assert_array_equal(aug.Invert(sr)(x), -x)
    np.testing.assert_array_equal(aug.Reverse(sr)(x), x[::-1])
    np.testing.assert_allclose(aug.Clipping(sr, threshold_db=-6)(x), orc.fx_clipping(x, -6), rtol=1e-6)
    np.testing.assert_allclose(aug.Distortion(sr, drive_db=12.0)(x), orc.fx_distortion(x, 12.0), atol=2e-6)
    np.testing.assert_allclose(aug.Bitcrush(sr, bit_depth=8)(x), orc.fx_bitcrush(x, 8), atol=1e-7)
    assert rel_rms(aug.Preemphasis(sr, coef=0.97)(x), orc.fx_preemphasis(x, 0.97)) < 1e-6
    assert rel_rms(aug.Deemphasis(sr, coef=0.9)(x), orc.fx_deemphasis(x, 0.9)) < 1e-5
    # the two emphasis filters invert each other (librosa's design)
    assert rel_rms(aug.Deemphasis(sr, coef=0.5)(aug.Preemphasis(sr, coef=0.5)(x)), x) < 1e-5
    for shape in aug.Fade.FADE_SHAPES:
        f = aug.Fade(sr, fade_in_len=0.3, fade_out_len=0.2, fade_in_shape=shape, fade_out_shape=shape)
        assert rel_rms(f(x), orc.fx_fade(x.astype(np.float64), sr, 0.3, 0.2, shape, shape)) < 1e-5
    # reference tests/test_augmentation.py:300-327 and 504-515
    y = aug.Fade(sr, 0.25, 0.25, "linear", "linear")(np.ones(n, dtype=np.float32))
    assert abs(y[0]) < 1e-6 and abs(y[-1]) < 1e-6
    np.testing.assert_array_equal(aug.Invert(sr)(np.ones(16, dtype=np.float32)), -np.ones(16))
    assert rel_rms(aug.peak_normalize(x), orc.peak_normalise_clip(x)) < 1e-6


@pytest.mark.parametrize("cls,mode", [(aug.TimeWarpSilence, "silence"), (aug.TimeWarpDuplicate, "duplicate"),
                                      (aug.TimeWarpRemove, "remove"), (aug.TimeWarpReverse, "reverse")])
@pytest.mark.parametrize("n,fps", [(8000, 7.0), (3000, 2.0)])
def test_timewarp_matches_reference_semantics(cls, mode, n, fps):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n).astype(np.float32)
    sr = 8000
    fx = cls(sr, fps=fps, prob=0.3)
    random.seed(11)
    got = fx(x)
    random.seed(11)
    fl = round(sr / fps)
    decisions = [random.random() < 0.3 for _ in range(1 if fl > n else fl)]
    want = orc.fx_wrap(lambda a: orc.fx_timewarp(a, sr, fps, decisions, mode), x)
    assert got.shape == x.shape
    np.testing.assert_allclose(got, want, atol=0)


def test_event_fx_chain_and_dict_roundtrip():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(2000).astype(np.float32)
    chain = [aug.Gain(8000, gain_db=3.0), aug.Invert(8000)]
    ev = core.Event("fx", x, 8000, augmentations=chain)
    want = orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(x, 3.0)))
    assert rel_rms(ev.load_audio(), want) < 1e-6
    assert ev.load_audio() is ev.audio                      # cached (event.py:511-512)
    ev.register_augmentations(aug.Reverse(8000))
    assert ev.audio is None                                 # registering FX invalidates caches (event.py:739-782)
    d = chain[0].to_dict()
    assert d == dict(name="Gain", sample_rate=8000, gain_db=3.0)
    assert aug.Augmentation.from_dict(d) == chain[0]


@pytest.mark.parametrize("beta", [0, 1, 2, -1])
@pytest.mark.parametrize("n", [1000, 1001])   # 1000: mixed radix 2/5; 1001 = 7*11*13: Bluestein
def test_powerlaw_noise_matches_reference(golden, beta, n):
    from audiblelight_amd import ambience as amb

    got = amb.powerlaw_psd_gaussian(beta, (4, n))
    assert got.shape == (4, n)
    assert rel_rms(got, golden[f"g6_b{beta}_n{n}"]) < 2e-6


def test_powerlaw_misc_and_ambience_class(golden):
    from audiblelight_amd import ambience as amb

    assert rel_rms(amb.powerlaw_psd_gaussian(1, (2, 512), fmin=0.1, seed=7), golden["g6_fmin"]) < 2e-6
    assert rel_rms(amb.powerlaw_psd_gaussian(1, 300), golden["g6_1d"]) < 2e-6
    with pytest.raises(ValueError, match="fmin"):
        amb.powerlaw_psd_gaussian(1, 16, fmin=0.7)
    a = amb.Ambience(channels=4, duration=0.5, alias="amb", noise="pink", ref_db=-60, sample_rate=8000)
    noise = a.load_ambience()
    assert rel_rms(noise, golden["g7_noise"]) < 2e-6
    np.testing.assert_allclose(np.abs(noise).max(axis=1), 1.0, rtol=1e-6)   # reference tests/test_ambience.py:134-136
    assert a.load_ambience() is noise                                       # cached
    assert amb.Ambience.from_dict(a.to_dict()) == a
    with pytest.raises(AttributeError):
        amb.Ambience(channels=1, duration=1, alias="x")
    with pytest.raises(KeyError):
        amb.Ambience(channels=1, duration=1, alias="x", noise="mauve")
    # file-style ambience: a mono clip tiled over channels and time (reference ambience.py:172-208)
    clip = np.arange(1, 301, dtype=np.float32) / 300
    t = amb.Ambience(channels=3, duration=0.1, alias="t", clip=clip, sample_rate=8000).load_ambience()
    np.testing.assert_allclose(t, orc.peak_normalise_rows(orc.tile_ambience(clip, 3, 800).astype(np.float64)), rtol=1e-6)


def test_scene_with_device_ambience(golden):
    from audiblelight_amd import ambience as amb

    scene = build_g8_scene(golden, with_ambience=False)
    scene.add_ambience(amb.Ambience(channels=4, duration=2.0, alias="a0", noise="white", ref_db=-65, sample_rate=8000))
    out = scene.generate()["mic000"]
    assert_parity(out, golden["g8_scene"], TOL)
    with pytest.raises(ValueError, match="does not match expected shape"):
        bad = build_g8_scene(golden, with_ambience=False)
        bad.add_ambience(amb.Ambience(channels=4, duration=1.0, alias="a0", noise="white", sample_rate=8000))
        bad.generate()


def test_two_microphones_with_different_capsule_counts(golden):
    """render_audio_for_all_scene_events loops microphones (synthesize.py:653-675): one batch per microphone."""
    rng = np.random.default_rng(2)
    sr = 8000
    irs = {"mic000": rng.standard_normal((4, 3, 300)), "mic001": rng.standard_normal((2, 3, 300))}
    scene = core.Scene(1.0, core.StaticIRState(irs), sample_rate=sr, ref_db=-60)
    clips = [rng.standard_normal(n).astype(np.float32) for n in (2000, 1500)]
    scene.add_event(core.Event("a", clips[0], sr, snr=10.0, scene_start=0.1, n_emitters=1))
    scene.add_event(core.Event("b", clips[1], sr, snr=20.0, scene_start=0.5, n_emitters=2))
    out = scene.generate()
    assert out["mic000"].shape == (4, 8000) and out["mic001"].shape == (2, 8000)
    for mic, h in irs.items():
        ev_a, ev_b = scene.events["a"], scene.events["b"]
        want_a = orc.render_event(ev_a.load_audio(), h[:, :1, :], 10.0, ref_db=-60, sr=sr)["spatial"]
        want_b = orc.render_event(ev_b.load_audio(), h[:, 1:3, :], 20.0, ref_db=-60, is_moving=True, duration=ev_b.duration, sr=sr)["spatial"]
        assert_parity(ev_a.spatial_audio[mic], want_a, TOL)
        assert_parity(ev_b.spatial_audio[mic], want_b, TOL)
        ref = orc.mix_scene([want_a, want_b], [(0.1, ev_a.scene_end), (0.5, ev_b.scene_end)], 1.0, sr, keep_padded=False)["scene"]
        assert_parity(out[mic], ref, TOL)


def test_scene_json_round_trip(golden, tmp_path):
    """Metadata round trip (reference core.py:2106-2243): a scene rebuilt from JSON + arrays renders identically."""
    scene = build_g8_scene(golden, with_ambience=False)
    scene.events["ev0"].register_augmentations([aug.Gain(8000, gain_db=-2.0), aug.Invert(8000)])
    from audiblelight_amd import ambience as amb

    scene.add_ambience(amb.Ambience(channels=4, duration=2.0, alias="a0", noise="white", ref_db=-65, sample_rate=8000))
    first = scene.generate()["mic000"].copy()
    path = str(tmp_path / "scene.json")
    scene.to_json(path)
    again = core.Scene.from_json(path, clips={a: e._raw for a, e in scene.events.items()}, irs=dict(scene.state.irs))
    strip = lambda d: {k: v for k, v in d.items() if k != "creation_time"}   # noqa: E731
    assert strip(again.to_dict()) == strip(scene.to_dict())
    np.testing.assert_array_equal(again.generate()["mic000"], first)


def test_reference_format_scene_json_renders_like_the_reference(tmp_path):
    """metadata written by the REFERENCE's own Scene/Event/Ambience/MicArray ``to_dict`` code
    (tests/golden/reference_scene.json, made by tests/golden/make_scene_json.py; core.py:2106-2130, event.py:568-620,
    ambience.py:219-233) + the arrays it does not carry -> ``Scene.from_json`` -> ``generate()`` equals what the
    reference rendered for that scene: static, FX-chained, moving and dry-path events, ambience, two microphones."""
    import json
    import os

    here = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(here, "reference_scene_arrays.npz"))
    meta = json.load(open(os.path.join(here, "reference_scene.json")))
    clips = {a: z[f"clip_{a}"] for a in meta["events"]}
    irs = {m: z[f"irs_{m}"] for m in meta["state"]["microphones"]}
    scene = core.Scene.from_json(os.path.join(here, "reference_scene.json"), clips, irs)
    assert [len(e) for e in scene.events.values()] == [1, 1, 4, 1] and scene.events["event002"].is_moving
    assert [type(a).__name__ for a in scene.events["event001"].augmentations] == ["Gain", "Invert"]
    out = scene.generate(output_dir=str(tmp_path), audio_subtype="FLOAT")
    for mic in irs:
        assert_parity(out[mic], z[f"scene_{mic}"], TOL)
        for alias, ev in scene.events.items():
            assert_parity(ev.spatial_audio[mic], z[f"spatial_{mic}_{alias}"], TOL)
    assert_parity(scene.events["event003"]._spatial_audio_dry["mic000"], z["dry_mic000_event003"], TOL)
    # what we write back has the reference's layout: same keys at every level the reference's from_dict reads
    ours = json.load(open(tmp_path / "metadata_out.json"))
    assert set(ours) == set(meta)
    for alias in meta["events"]:
        assert set(ours["events"][alias]) == set(meta["events"][alias])
        for k in ("scene_start", "scene_end", "duration", "snr", "num_emitters", "augmentations", "emitters", "class_label",
                  "filepath", "event_start", "is_moving", "ref_ir_channel", "direct_path_time_ms"):
            assert ours["events"][alias][k] == meta["events"][alias][k], (alias, k)
    assert ours["ambience"] == meta["ambience"]
    assert ours["state"]["microphones"] == meta["state"]["microphones"] and ours["state"]["emitters"] == meta["state"]["emitters"]
    with pytest.raises(KeyError, match="No clip given"):
        core.Scene.from_dict(meta, {}, irs)
    with pytest.raises(ValueError, match="emitter columns"):
        core.Scene.from_dict(meta, clips, {m: v[:, :3] for m, v in irs.items()})


def test_scene_generate_argument_list(golden, tmp_path):
    """Scene.generate takes the reference's arguments (core.py:1789-1799): WAV per microphone + metadata JSON on disk;
    the host-side outputs that are out of scope raise."""
    from scipy.io import wavfile

    scene = build_g8_scene(golden, with_ambience=False)
    out = scene.generate(output_dir=str(tmp_path), audio_fname="mix.wav", metadata_fname="meta", metadata_dcase=False)
    sr, data = wavfile.read(str(tmp_path / "mix_mic000.wav"))
    assert sr == scene.sample_rate and data.dtype == np.int16       # soundfile's default subtype (core.py:1840-1847)
    np.testing.assert_array_equal(data, pcm16(out["mic000"].T))
    scene.generate(output_dir=str(tmp_path), audio_fname="mixf", audio_subtype="FLOAT", metadata_dcase=False)
    sr, data = wavfile.read(str(tmp_path / "mixf_mic000.wav"))
    assert data.dtype == np.float32
    np.testing.assert_array_equal(data.T, out["mic000"])
    assert (tmp_path / "meta.json").exists()
    with pytest.raises(ValueError, match="DCASE"):        # events built from bare arrays carry no class index / positions,
        scene.generate(output_dir=str(tmp_path), audio=False)   # and DCASE metadata is ON by default (core.py:1794)
    with pytest.raises(NotImplementedError):
        scene.generate(video=True)


def test_stft_helpers_match_reference():
    """stft / perform_time_variant_convolution / istft_overlap_synthesis with the reference's signatures
    (synthesize.py:109-145,184-274), device kernels behind them, against the reference's own outputs (G10)."""
    import os

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_stft_vectors.npz"))
    a, h = z["g10_audio"], z["g10_irs"]

    def close(got, want, tol=2e-5):
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= tol * np.abs(want).max()

    s_a = syn.stft(a.astype(np.float64))
    assert s_a.dtype == np.complex128 and s_a.flags.c_contiguous
    close(s_a, z["g10_stft_audio"])
    s_h = syn.stft(h)
    close(s_h, z["g10_stft_irs"])
    close(syn.stft(h, stft_dims_first=False), z["g10_stft_irs_dims_last"])
    close(syn.stft(a[:777], 256, 128, 64), z["g10_stft_b"])
    y = syn.perform_time_variant_convolution(z["g10_stft_audio"], z["g10_stft_irs"], z["g10_w"])
    close(y, z["g10_tv"])
    x = syn.istft_overlap_synthesis(z["g10_tv"], 512, 256, 128)
    assert_parity(x, z["g10_istft"], TOL)
    assert_parity(syn.istft_overlap_synthesis(z["g10_stft_b"][:, :, None], 256, 128, 64), z["g10_istft_b"], TOL)
    # the chain of the three equals time_variant_convolution's envelope-form render of the same event
    ev = core.Event("mv", a, 8000, snr=10.0, n_emitters=3)
    tv = syn.time_variant_convolution(h.astype(np.float64), ev)
    chain = syn.istft_overlap_synthesis(syn.perform_time_variant_convolution(s_a, s_h, z["g10_w"])).T
    assert_parity(tv[:, : chain.shape[1]], chain[:, : tv.shape[1]], TOL)
    # an fft size with prime factors above 7 goes through Bluestein's chirp-z (it was refused until round 6): against numpy's rfft
    big = syn.stft(a, 2 * 11 * 13, 256, 128)
    ref = orc.stft_frames(a.astype(np.float64), 2 * 11 * 13, 256, 128)
    assert big.shape == ref.shape
    assert_parity(big.real, ref.real, 1e-5, what="stft 286 re")
    assert_parity(big.imag, ref.imag, 1e-5, what="stft 286 im")


def test_moving_events_under_other_stft_geometries():
    """A7 with fft_size / win_size / hop_size other than the defaults (synthesize.py:507-516 exposes them, :277-310 honours
    any geometry the framing accepts) against the reference's own renders (G14, tests/golden/make_golden.py::geometry_vectors):
    render_event_audio (normalize_irs -> time-variant convolution -> truncation -> level law) and time_variant_convolution,
    through the envelope form where it holds (the scaled default 1024/512/256) and through the device STFT chain elsewhere
    (win != 2*hop, fft < 2*win - 1, 75 % overlap, non-power-of-two sizes); what the reference refuses raises ValueError here."""
    import os

    with np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_geometry_vectors.npz")) as z:
        z = {k: z[k] for k in z.files}
    a, h = z["g14_audio"], z["g14_irs"]
    r = syn.get_renderer()
    for fft_size, win, hop in z["g14_geometries"].tolist():
        tag = f"g14_{fft_size}_{win}_{hop}"
        ev = core.Event("g14", a, 8000, snr=11.0, n_emitters=4, is_moving=True)
        syn.render_event_audio(ev, h, "mic000", ref_db=-65, fft_size=fft_size, win_size=win, hop_size=hop)
        got = ev.spatial_audio["mic000"]
        assert got.dtype == np.float64 and got.shape == z[tag + "_spatial"].shape
        assert_parity(got, z[tag + "_spatial"], TOL, what=tag)
        hn = syn.normalize_irs(h.astype(np.float64).transpose(1, 0, 2)).transpose(1, 0, 2)
        raw = syn.time_variant_convolution(hn, ev, fft_size, win, hop)
        assert raw.shape == z[tag + "_raw"].shape
        assert_parity(raw, z[tag + "_raw"], TOL, what=tag)
    # an IR whose 4-float padded row (773 -> 776 samples) would cross a 2*hop boundary: the IR spectrogram's frame count must come
    # from the TRUE length (found by profiles/tools/fuzz_geometry.py, seed 335); against the oracle's literal restatement
    rng = np.random.default_rng(335)
    a2 = rng.standard_normal(7535).astype(np.float32)
    a2 /= np.abs(a2).max()
    h2 = (rng.standard_normal((2, 3, 773)) * np.exp(-np.arange(773) / 150.0)).astype(np.float32)
    ev = core.Event("odd", a2, 16000, snr=9.0, n_emitters=3, is_moving=True)
    syn.render_event_audio(ev, h2, "mic000", ref_db=-60, fft_size=512, win_size=477, hop_size=129)
    want = orc.render_event(a2, h2.astype(np.float64), 9.0, ref_db=-60, is_moving=True, duration=7535 / 16000, sr=16000, nfft=512, win=477,
                            hop=129)["spatial"]
    assert_parity(ev.spatial_audio["mic000"], want, TOL, what="odd IR length")
    # clips that reach the renderer with work still to do ON THE DEVICE -- folded scalar FX + peak normalisation (al_clip_scales)
    # and a device-resident FX chain -- go through the general chain too: the clip the STFT frames is the finished one
    raw = rng.standard_normal(5000).astype(np.float32) * 0.4
    for chain, finished in (([aug.Gain(16000, gain_db=-3.0), aug.Invert(16000)], orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(raw, -3.0)))),
                            ([aug.Reverse(16000), aug.Gain(16000, gain_db=2.0)], orc.peak_normalise_clip(orc.fx_gain(orc.fx_reverse(raw), 2.0)))):
        ev = core.Event("fx", raw, 16000, snr=12.0, n_emitters=3, is_moving=True, augmentations=chain)
        syn.render_event_audio(ev, h2, "mic000", ref_db=-60, fft_size=384, win_size=256, hop_size=64)
        want = orc.render_event(finished, h2.astype(np.float64), 12.0, ref_db=-60, is_moving=True, duration=5000 / 16000, sr=16000,
                                nfft=384, win=256, hop=64)["spatial"]
        assert_parity(ev.spatial_audio["mic000"], want, TOL, what="fx chain + general geometry")
    # the mixdown takes such an event like any other (same RenderResult): scaled render summed into a scene
    ev = core.Event("g14", a, 8000, snr=11.0, n_emitters=4, is_moving=True, scene_start=0.1)
    syn.render_event_audio(ev, h, "mic000", ref_db=-65, fft_size=512, win_size=256, hop_size=192)
    assert ev.spatial_audio.is_resident("mic000")
    for fft_size, win, hop in z["g14_refused"].tolist():
        ev = core.Event("g14", a, 8000, snr=11.0, n_emitters=4, is_moving=True)
        with pytest.raises(ValueError):
            syn.render_event_audio(ev, h, "mic000", ref_db=-65, fft_size=fft_size, win_size=win, hop_size=hop)
        with pytest.raises(ValueError):
            syn.time_variant_convolution(h.astype(np.float64), ev, fft_size, win, hop)
    # a static event ignores the geometry arguments entirely, as in the reference (it never frames anything)
    ev = core.Event("s", a, 8000, snr=11.0)
    syn.render_event_audio(ev, h[:, :1], "mic000", ref_db=-65, fft_size=1024, win_size=512, hop_size=128)
    assert np.isfinite(ev.spatial_audio["mic000"]).all()
    assert r is syn.get_renderer()


def test_general_stft_path_limits():
    """Advisor r05: the literal STFT chain behind non-default geometries has no limits of its own left.  More output frames than one
    launch's grid holds (hop 16 at 44.1 kHz passes 65 535 frames after 23 s of clip): al_tv_stft_mac walks the frame axis in groups --
    70 000 frames here against the defining sum (synthesize.py:217-250).  An fft size with a prime factor above 7 (the reference's
    numpy rfft / irfft take any size, synthesize.py:135,263): al_stft / al_istft_ola go through Bluestein's chirp-z -- fft sizes 22
    (2 x 11), 23 (prime, odd) and 26 (2 x 13) against the oracle's literal STFT-domain restatement, every row, through
    render_event_audio and time_variant_convolution alike."""
    rng = np.random.default_rng(5)
    n_frames, f_ir, n_freq, n_ch, n_irs = 70_000, 2, 2, 1, 2
    s_a = (rng.standard_normal((n_frames, n_freq)) + 1j * rng.standard_normal((n_frames, n_freq))).astype(np.complex64)
    s_ir = (rng.standard_normal((f_ir, n_freq, n_ch, n_irs)) + 1j * rng.standard_normal((f_ir, n_freq, n_ch, n_irs))).astype(np.complex64)
    w = rng.random((n_frames, n_irs)).astype(np.float32)
    got = syn.perform_time_variant_convolution(s_a, s_ir, w)
    ctf = np.einsum("il,kfcl->ikfc", w.astype(np.float64), s_ir.astype(np.complex128))        # (i, k, f, c)
    want = s_a[:, :, None] * ctf[:, 0]
    want[1:] += s_a[:-1, :, None] * ctf[:-1, 1]
    assert got.shape == (n_frames, n_freq, n_ch)
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max() and np.abs(got[-1]).max() > 0      # the last group was written too

    import types

    a = rng.standard_normal(3000).astype(np.float32)
    a /= np.abs(a).max()
    h = (rng.standard_normal((2, 3, 200)) * np.exp(-np.arange(200) / 40.0)).astype(np.float32)
    for fft, win, hop in ((22, 12, 5), (23, 12, 6), (26, 13, 7)):
        ev = types.SimpleNamespace(alias="m", snr=7.0, sample_rate=8000, is_moving=True, duration=3000 / 8000, spatial_audio={},
                                   _spatial_audio_dry={}, ref_ir_channel=None, direct_path_time_ms=None,
                                   load_audio=lambda ignore_cache=False, normalize=True: a, __len__=lambda: 3)
        syn.render_event_audio(ev, h, "m", ref_db=-60, fft_size=fft, win_size=win, hop_size=hop)
        want = orc.render_event(a, h.astype(np.float64), 7.0, ref_db=-60, is_moving=True, duration=3000 / 8000, sr=8000, nfft=fft, win=win,
                                hop=hop)["spatial"]
        assert_parity(ev.spatial_audio["m"], want, TOL, what=(fft, win, hop))
        raw = syn.time_variant_convolution(h, ev, fft_size=fft, win_size=win, hop_size=hop)
        assert_parity(raw, orc.convolve_moving_stft(a, h.astype(np.float64), 3000 / 8000, 8000, fft, win, hop), TOL, what=("raw", fft))
    # the forward helper itself at a prime size, against the oracle's frames (numpy rfft)
    y = rng.standard_normal((2, 500)).astype(np.float32)
    spec, ref = syn.stft(y, fft_size=23, win_size=12, hop_size=6), orc.stft_frames(y.astype(np.float64), 23, 12, 6)
    assert spec.shape == ref.shape
    assert_parity(spec.real, ref.real, 1e-5, what="stft 23 re")
    assert_parity(spec.imag, ref.imag, 1e-5, what="stft 23 im")


def test_degenerate_events_render_like_the_reference():
    """Silence stays silence (G15: the reference's own renders of degenerate events): snr = 0, a negative snr, an all-zero IR, an
    all-zero clip, a moving event with one all-zero IR among its emitters, the dry render of an all-zero clip.  The reference forms
    1 / tiny and 10^(dB/20) / tiny in float64 and multiplies zeros by them; the device keeps every such scalar finite in float32
    (csrc/al_kernels.hip: finite_f32, emitter_gain_of).  Then the same events in ONE scene with a silent ambience and a silent clip
    under a +12 dB gain: the mix is finite and equals the oracle's."""
    import os

    with np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_edge_vectors.npz")) as z:
        z = {k: z[k] for k in z.files}
    a, h, h3 = z["g15_audio"], z["g15_irs"], z["g15_irs_moving"]
    cases = {"snr0": (a, h, 0.0, {}), "snr_neg": (a, h, -4.0, {}), "zero_ir": (a, np.zeros_like(h), 9.0, {}),
             "zero_clip": (np.zeros_like(a), h, 9.0, {}), "moving_one_zero_ir": (a, h3, 7.0, dict(n_emitters=3, is_moving=True)),
             "zero_clip_dry": (np.zeros_like(a), h, 9.0, dict(ref_ir_channel=0, direct_path_time_ms=[2, 20]))}
    for tag, (clip, irs, snr, kw) in cases.items():
        ev = core.Event(tag, clip, 8000, snr=snr, **kw)
        ev.audio = np.asarray(clip, dtype=np.float32)       # the finished clip, as the golden's event hands it over (no peak normalisation of zeros)
        syn.render_event_audio(ev, irs, "mic000", ref_db=-65)
        got, want = ev.spatial_audio["mic000"], z[f"g15_{tag}_spatial"]
        assert np.isfinite(got).all() and got.shape == want.shape, tag
        if np.abs(want).max() == 0:
            assert np.abs(got).max() == 0, tag
        else:
            assert_parity(got, want, TOL, what=tag)
        if "ref_ir_channel" in kw:
            dry = ev._spatial_audio_dry["mic000"]
            assert np.isfinite(dry).all() and np.abs(dry).max() == 0
    # ... while non-finite INPUT still fails the reference's finite check (librosa.util.valid_audio of the render, synthesize.py:603):
    # the saturating scalars above must not launder a NaN or an Inf in an IR into silence
    for poison in (np.nan, np.inf):
        bad_ir = h.copy()
        bad_ir[1, 0, 17] = poison
        with pytest.raises(ValueError, match="not finite"):
            syn.render_event_audio(core.Event("poisoned", a, 8000, snr=9.0), bad_ir, "mic000", ref_db=-65)
        bad3 = h3.copy()
        bad3[0, 2, 5] = poison
        with pytest.raises(ValueError, match="not finite"):
            syn.render_event_audio(core.Event("poisoned", a, 8000, snr=9.0, n_emitters=3, is_moving=True), bad3, "mic000", ref_db=-65)
    # one scene: a normal event, a silent clip under Gain(+12 dB) (folded scalar: 4 / tiny32 would overflow float32), an event
    # whose IR is all zeros, and an ambience of silence; scene.audio is finite and equals the oracle's mix of the one audible event
    rng = np.random.default_rng(15)
    sr, C, L = 8000, 2, 400
    irs = np.concatenate([h, h, np.zeros_like(h)], axis=1)
    scene = core.Scene(0.5, core.StaticIRState({"mic000": irs}), sample_rate=sr, ref_db=-65)
    scene.add_event(core.Event("loud", a, sr, snr=10.0, scene_start=0.05))
    scene.add_event(core.Event("silent", np.zeros(1200, np.float32), sr, snr=10.0, scene_start=0.1, augmentations=[aug.Gain(sr, gain_db=12.5)]))
    scene.add_event(core.Event("deaf", a[:1000], sr, snr=10.0, scene_start=0.2))
    from audiblelight_amd import ambience as amb
    scene.add_ambience(amb.Ambience(C, 0.5, alias="hush", clip=np.zeros((1, 4000), np.float32), ref_db=-60, sample_rate=sr))
    got = scene.generate()["mic000"]
    assert np.isfinite(got).all()
    want = orc.render_event(orc.peak_normalise_clip(a), h.astype(np.float64), 10.0, ref_db=-65, sr=sr)["spatial"]
    ref = orc.mix_scene([want], [(0.05, 0.05 + len(a) / sr)], 0.5, sr, keep_padded=False)["scene"]
    assert_parity(got, ref, TOL)
    for ev in scene.events.values():
        assert np.isfinite(ev.spatial_audio["mic000"]).all()


def test_fx_chain_stays_on_device_and_scalars_fold():
    """BASELINE configs[4]'s "gain/polarity augmentations fused" through the PRODUCT classes: events built with
    ``augmentations=[Gain, Invert]`` render to what the oracle gives for peak_normalise(invert(gain(raw))) with no FX
    kernel and no clip statistic crossing PCIe (the scalar is evaluated on the device and folded into the spectra);
    a chain with non-scalar FX runs on ONE device-resident clip and is handed to the renderer in HBM (event.py:529-539,
    augmentation.py:91-136)."""
    rng = np.random.default_rng(11)
    sr, C, L = 8000, 3, 700
    raws = [rng.standard_normal(n).astype(np.float32) * s for n, s in ((3000, 0.3), (2500, 2.0), (2800, 1.0))]
    irs = (rng.standard_normal((C, 3, L)) * np.exp(-np.arange(L) / 150.0)).astype(np.float32)
    chains = [[aug.Gain(sr, gain_db=-4.0), aug.Invert(sr)],                       # pure scalars: folded
              [aug.Invert(sr), aug.Gain(sr, gain_db=7.5), aug.Invert(sr)],         # folded, net positive
              [aug.Fade(sr, 0.05, 0.1, "linear", "half_sine"), aug.Gain(sr, gain_db=2.0), aug.Reverse(sr)]]   # device chain
    scene = core.Scene(1.0, core.StaticIRState({"mic000": irs}), sample_rate=sr, ref_db=-65)
    for i, (raw, chain) in enumerate(zip(raws, chains)):
        scene.add_event(core.Event(f"e{i}", raw, sr, snr=10.0 + i, scene_start=0.1 * i, augmentations=chain))
    assert aug.fold_scalars(chains[0]) == pytest.approx(-(10 ** (-4.0 / 20)), rel=1e-6) and aug.fold_scalars(chains[2]) is None
    scene.generate()
    want_clips = [orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(raws[0], -4.0))),
                  orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(orc.fx_invert(raws[1]), 7.5))),
                  orc.peak_normalise_clip(orc.fx_reverse(orc.fx_gain(orc.fx_fade(raws[2], sr, 0.05, 0.1, "linear", "half_sine"), 2.0)))]
    spatials = []
    for i, ev in enumerate(scene.events.values()):
        want = orc.render_event(want_clips[i], irs[:, [i], :].astype(np.float64), ev.snr, sr=sr)["spatial"]
        spatials.append(want)
        assert_parity(ev.spatial_audio["mic000"], want, TOL)
        assert ev.audio is None                         # nobody asked for the host clip: it never came back
    assert getattr(scene.events["e0"], "_last_chain", None) is None          # folded: no FX kernel, no device clip
    chain_clip = scene.events["e2"]._last_chain
    assert chain_clip.uploads == 1 and chain_clip.downloads == 0            # three FX + normalisation, zero D2H
    ref = orc.mix_scene(spatials, [(e.scene_start, e.scene_end) for e in scene.events.values()], 1.0, sr, keep_padded=False)["scene"]
    assert_parity(scene.audio["mic000"], ref, TOL)
    # the host API still gives the reference's answer, through one upload and one download
    ev = core.Event("h", raws[2], sr, augmentations=chains[2])
    assert rel_rms(ev.load_audio(), want_clips[2]) < 1e-6
    assert ev._last_chain.uploads == 1 and ev._last_chain.downloads == 1


def test_ir_ingest_ragged_packing_and_resampling():
    """(i) ragged per-(capsule, source) IRs -> the zero-padded float32 HBM layout, equal to the
    reference's zero_arr fill (worldstate.py:2213-2253), and rendered straight from that buffer; (ii) IRs at another
    sample rate resampled on the device, pinned to scipy.signal.resample_poly (the reference's librosa/soxr resampler
    is an absent third-party algorithm: unpinned by definition)."""
    from scipy.signal import resample_poly

    from audiblelight_amd import ingest, plan as planning

    r = syn.get_renderer()   # the module fixture's renderer (host emulation here, the gfx950 build in test_gpu_api.py)
    rng = np.random.default_rng(5)
    C, N = 3, 2
    lens = rng.integers(0, 700, size=(C, N))
    lens[1, 1] = 0                                         # a silent path
    nest = [[(rng.standard_normal(lens[c, n]) * np.exp(-np.arange(lens[c, n]) / 120.0)).astype(np.float64 if (c + n) % 2 else np.float32)
             for n in range(N)] for c in range(C)]
    dev, strides, maxlen = ingest.pack_ragged_irs(r, nest)
    want = np.zeros((C, N, maxlen))                        # the reference's zero_arr, filled the reference's way
    for c in range(C):
        for n in range(N):
            want[c, n, : lens[c, n]] = nest[c][n]
    pitch = strides[1]
    got = r.mem.download(dev)[: C * N * pitch].reshape(C, N, pitch)
    np.testing.assert_array_equal(got[:, :, :maxlen], want.astype(np.float32))
    assert not got[:, :, maxlen:].any() and maxlen == lens.max() and strides == (N * pitch, pitch)
    clips = [rng.standard_normal(1500).astype(np.float32) for _ in range(N)]
    specs = [planning.EventSpec(n_samples=1500, n_emitters=1, snr=10.0, emitter0=n) for n in range(N)]
    pl = planning.plan_batch(specs, C, maxlen, 8000, log2_block=10)
    res = r.prepare(pl, clips, dev, strides).run()
    for n in range(N):
        ref = orc.render_event(clips[n], want[:, [n], :], 10.0, sr=8000)["spatial"]
        assert_parity(res.spatial_audio(n), ref, TOL)
    # resampling 44.1 kHz -> 48 kHz (160/147) and 48 -> 16 kHz (1/3)
    h = (rng.standard_normal((2, 3, 900)) * np.exp(-np.arange(900) / 150.0)).astype(np.float32)
    for a, b in ((44100, 48000), (48000, 16000)):
        got = ingest.resample_irs(r, h, a, b)
        up, down = (160, 147) if a == 44100 else (1, 3)
        ref = resample_poly(h.astype(np.float64), up, down, axis=-1)
        n_out = int(round(900 * b / a))
        assert got.shape == (2, 3, n_out) and got.dtype == np.float32
        m = min(n_out, ref.shape[-1])
        assert rel_rms(got[..., :m], ref[..., :m]) < 1e-5
    assert ingest.resample_irs(r, h, 48000, 48000) is not None


def test_encode_frames_every_store_path():
    """al_encode_frames: (C, T) float32 scene -> (T, C) interleaved WAV payload (core.py:1840-1847, soundfile's PCM_16
    default = lrint(x * 0x7FFF), saturated).  Capsule counts that take the 16-byte store paths (C % 8 == 0 for PCM_16,
    C % 4 == 0 for float) and ones that do not, lengths that end inside a 64-frame tile, values beyond full scale."""
    import ctypes as ct

    r = syn.get_renderer()
    rng = np.random.default_rng(11)
    for C, T in ((32, 1000), (8, 64), (40, 777), (4, 130), (12, 65), (5, 333), (1, 50)):
        scene = (rng.standard_normal((C, T)) * 0.6).astype(np.float32)
        scene[0, :4] = (1.5, -1.5, 1.0, -1.0)
        dev = r.mem.upload(scene.reshape(-1))
        for fmt, dtype in ((_hip.FRAMES_PCM16, np.int16), (_hip.FRAMES_F32, np.float32)):
            out = r.mem.zeros(C * T + 8, dtype)
            r.lib.call("al_encode_frames", r.mem.ptr(dev), C, T, fmt, r.mem.ptr(out), r.mem.stream())
            r.mem.synchronize()
            got = r.mem.download(out)
            assert not got[C * T:].any(), (C, T, fmt)                    # nothing past the payload
            got = got[: C * T].reshape(T, C)
            if dtype == np.int16:
                want = pcm16(scene.T)
            else:
                want = scene.T
            np.testing.assert_array_equal(got, want, err_msg=str((C, T, fmt)))


def test_event_from_wav_file_resamples_on_the_device(tmp_path):
    """an event read the way the reference reads one (librosa.load(path, sr, mono=True, offset,
    duration), event.py:520-527): excerpt of a stereo 44.1 kHz PCM_16 file, mono down-mix, resampled to the scene rate on
    the device as the first step of the FX chain (pinned to scipy.signal.resample_poly; librosa's soxr resampler is an
    absent third-party algorithm), then FX + peak normalisation, one upload and no host round trip before the render."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly

    from audiblelight_amd import ingest

    rng = np.random.default_rng(21)
    sr_file, sr = 44100, 48000
    t = np.arange(int(1.5 * sr_file)) / sr_file
    stereo = np.stack([0.5 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.standard_normal(len(t)),
                       0.3 * np.sin(2 * np.pi * 1000 * t)], axis=1)
    pcm = np.clip(np.rint(stereo * 32767), -32768, 32767).astype(np.int16)
    path = str(tmp_path / "clip.wav")
    wavfile.write(path, sr_file, pcm)
    offset, duration = 0.25, 0.8
    ev = core.Event.from_file(path, "event000", sr, event_start=offset, duration=duration, snr=12.0,
                              augmentations=[aug.Gain(gain_db=-3.0), aug.Invert()])
    assert ev.native_sample_rate == sr_file and ev.duration == duration and ev.filepath == path
    # the reference's steps on the host
    mono = (pcm[int(offset * sr_file): int(offset * sr_file) + int(duration * sr_file)].astype(np.float32) / 32768.0).mean(axis=1)
    np.testing.assert_allclose(ingest.read_wav_excerpt(path, offset, duration)[0], mono, atol=1e-7)
    n_out = int(np.ceil(len(mono) * sr / sr_file))
    res = resample_poly(mono.astype(np.float64), 160, 147)
    want = np.zeros(n_out)
    want[: min(n_out, len(res))] = res[:n_out]
    want = -want * 10 ** (-3.0 / 20)
    want = want / np.max(np.abs(want))
    got = ev.load_audio()
    assert got.shape == (n_out,) and got.dtype == np.float32
    assert rel_rms(got, want) < 1e-5
    assert ev._last_chain.uploads == 1 and ev._last_chain.downloads == 1
    # ... and handed to the renderer without coming back: the scalar chain is NOT folded (the raw clip is at another rate)
    ev.clear_audio()
    src = ev.clip_source()
    assert src.host is None and src.n == n_out
    # a file already at the scene rate goes the folded way (raw clip + one device scalar)
    wavfile.write(path, sr, pcm)
    ev2 = core.Event.from_file(path, "event001", sr, augmentations=[aug.Gain(gain_db=-3.0)])
    assert ev2.clip_source().host is not None and ev2.duration == pytest.approx(len(pcm) / sr)


def test_fx_match_the_reference_classes_outputs():
    """G11: the product's Fade / Invert / Reverse / TimeWarp* classes against what the reference's own classes returned for
    the same constructor arguments, input and Python random seed (tests/golden/make_fx_golden.py): float32 arithmetic on
    the device, so exact for the permutation FX and within float32 rounding of the float64 gain curves for Fade."""
    import os

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_fx_vectors.npz"))
    sr = int(z["sr"])
    for i, case in enumerate(z["fade_cases"]):
        a, b, la, lb, src = str(case).split(",")
        x = z["x_short" if src == "short" else "x"]
        got = aug.Fade(sample_rate=sr, fade_in_len=float(la), fade_out_len=float(lb), fade_in_shape=a, fade_out_shape=b)(x)
        assert got.shape == x.shape and got.dtype == z[f"fade_{i}"].dtype == np.float64    # float32 clip x float64 envelope
        np.testing.assert_allclose(got, z[f"fade_{i}"], rtol=0, atol=3e-7, err_msg=str(case))
    np.testing.assert_array_equal(aug.Invert(sample_rate=sr)(z["x"]), z["invert"])
    np.testing.assert_array_equal(aug.Reverse(sample_rate=sr)(z["x"]), z["reverse"])
    for i, case in enumerate(z["tw_cases"]):
        name, fps, prob, src, seed = str(case).split(",")
        fx = getattr(aug, name)(sample_rate=sr, fps=float(fps), prob=float(prob))
        random.seed(int(seed))
        got = fx(z[src])
        assert got.dtype == z[f"tw_{i}"].dtype, str(case)     # float64 where the reference spliced np.zeros(len(frame)) in
        np.testing.assert_array_equal(got, z[f"tw_{i}"], err_msg=str(case))


def test_ambience_file_mode_matches_the_reference(tmp_path):
    """G12: Ambience from a clip / file (ambience.py:170-214) against the reference's own load_ambience: mono tiled over
    channels and time, matching channel counts tiled over time only, a mismatched multichannel file reduced to one channel
    drawn with Python's random (same seed, same channel), truncation to round(duration * sr), per-channel peak
    normalisation.  Then the same from a PCM_16 WAV file at another sample rate (decode + device resampling)."""
    import os

    from scipy.io import wavfile
    from scipy.signal import resample_poly

    from audiblelight_amd import ambience as amb_mod

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_fx_vectors.npz"))
    sr, clip3 = int(z["sr"]), z["amb_clip3"]
    for j, case in enumerate(z["amb_cases"]):
        rows, channels, duration, seed = str(case).split(",")
        a = amb_mod.Ambience(channels=int(channels), duration=float(duration), alias=f"amb{j}", clip=clip3[: int(rows)], sample_rate=sr)
        random.seed(int(seed))
        got = a.load_ambience(ignore_cache=True, normalize=True)
        want = z[f"amb_{j}"]
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-7, err_msg=str(case))
        if z[f"amb_raw_{j}"].size:
            np.testing.assert_array_equal(a.load_ambience(ignore_cache=True, normalize=False), z[f"amb_raw_{j}"])
    # a stereo PCM_16 file at 6 kHz for an 8 kHz, 2-channel ambience
    pcm = np.clip(np.rint(clip3[:2].T * 0.5 * 32767), -32768, 32767).astype(np.int16)
    path = str(tmp_path / "amb.wav")
    wavfile.write(path, 6000, pcm)
    a = amb_mod.Ambience(channels=2, duration=0.7, alias="file", filepath=path, sample_rate=sr)
    got = a.load_ambience(normalize=False)
    src = resample_poly(pcm.T.astype(np.float64) / 32768.0, 4, 3, axis=-1)
    n_src = int(np.ceil(pcm.shape[0] * sr / 6000))
    src = np.pad(src, ((0, 0), (0, max(0, n_src - src.shape[1]))))[:, :n_src]
    want = np.tile(src, (1, -(-5600 // n_src)))[:, :5600]
    assert got.shape == (2, 5600) and rel_rms(got, want) < 1e-5


def test_big_batches_chunk_themselves(monkeypatch):
    """A batch whose spectra workspace exceeds the budget (AL_WORKSPACE_GB, default 40 % of the free HBM) is rendered as
    chunks of events over ONE reused workspace without the caller asking: same bits as the one-chunk render."""
    from audiblelight_amd import plan as planning

    r = syn.get_renderer()
    rng = np.random.default_rng(31)
    C, L, sr = 3, 2600, 8000
    clips = [rng.standard_normal(n).astype(np.float32) for n in (5000, 3000, 7001, 4100, 6000, 2000)]
    irs = (rng.standard_normal((C, len(clips), L)) * np.exp(-np.arange(L) / 400.0)).astype(np.float32)
    specs = [planning.EventSpec(n_samples=len(c), n_emitters=1, snr=12.0, emitter0=i) for i, c in enumerate(clips)]
    pl = planning.plan_batch(specs, C, L, sr, log2_block=10)
    set_switch(monkeypatch, "AL_WORKSPACE_GB", None)
    whole = r.prepare(pl, clips, irs)
    assert len(whole.descs) == 1
    want = [whole.run().spatial_audio(i).copy() for i in range(len(clips))]
    set_switch(monkeypatch, "AL_WORKSPACE_GB", str(pl.workspace_bytes() / 2.5 / 1e9))
    assert r.auto_chunk_events(pl) == 2
    chunked = r.prepare(pl, clips, irs)
    assert len(chunked.descs) == 3 and [d.n_events for d in chunked.descs] == [2, 2, 2]
    res = chunked.run()
    res.check_finite()
    for i in range(len(clips)):
        np.testing.assert_array_equal(res.spatial_audio(i), want[i])


def test_dcase_metadata_matches_the_reference_function(tmp_path):
    """G13: generate_dcase2024_metadata (host bookkeeping, synthesize.py:742-878) against the rows the reference's own function
    produced for the same events (two microphones, static and moving events, a shared audio file, two classes, an event clipped
    by the scene end); then through Scene.generate(metadata_dcase=True) on the reference-format scene, one CSV per microphone."""
    import json
    import os
    import types

    import pandas as pd

    here = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(here, "reference_fx_vectors.npz"))
    spec = json.loads(str(z["dcase_spec"]))
    mics = ["mic000", "mic001"]
    events = []
    for alias, cid, fname, t0, t1, polars in spec:
        rel = {m: [[az + 3.0 * k, el - 1.0 * k, d * (1 + 0.1 * k)] for az, el, d in polars] for k, m in enumerate(mics)}
        events.append(types.SimpleNamespace(alias=alias, class_id=cid, filename=fname, scene_start=t0, scene_end=t1,
                                            is_moving=len(polars) > 1, emitters_relative=rel))
    scene = types.SimpleNamespace(duration=10.0, state=types.SimpleNamespace(microphones={m: None for m in mics}),
                                  events={e.alias: e for e in events})
    got = syn.generate_dcase2024_metadata(scene)
    for m in mics:
        assert list(got[m].reset_index().columns) == list(z["dcase_columns"])
        np.testing.assert_array_equal(got[m].reset_index().to_numpy().astype(np.int64), z[f"dcase_{m}"])
    # the reference-format scene carries class indices and positions: CSVs next to the JSON
    arrays = np.load(os.path.join(here, "reference_scene_arrays.npz"))
    meta = json.load(open(os.path.join(here, "reference_scene.json")))
    ref_scene = core.Scene.from_dict(meta, {a: arrays[f"clip_{a}"] for a in meta["events"]},
                                     {m: arrays[f"irs_{m}"] for m in meta["state"]["microphones"]})
    ref_scene.generate(output_dir=str(tmp_path), audio=False, metadata_dcase=True)
    want = syn.generate_dcase2024_metadata(ref_scene)
    for m in meta["state"]["microphones"]:
        df = pd.read_csv(tmp_path / f"metadata_out_{m}.csv", header=None)
        np.testing.assert_array_equal(df.to_numpy(), want[m].reset_index().to_numpy())
        assert len(df) > 0 and set(df[1]) <= {e["class_id"] for e in meta["events"].values()}


def test_one_fx_realisation_per_event_across_microphones():
    """TimeWarp* draws its coin flips when it is applied (augmentation.py:1604-1790).  The reference's load_audio caches the
    clip, so every microphone, the dry path and a later load_audio see ONE realisation (event.py:507-510,538); here the
    chain runs once per event on the device and that resident clip serves them all."""
    rng = np.random.default_rng(77)
    sr = 8000
    raw = rng.standard_normal(4000).astype(np.float32)
    irs = {"mic_a": np.zeros((2, 1, 8), np.float32), "mic_b": np.zeros((3, 1, 8), np.float32)}
    irs["mic_a"][:, 0, 0] = 1.0      # unit impulses: the render is the (scaled) clip itself
    irs["mic_b"][:, 0, 0] = 1.0
    scene = core.Scene(1.0, core.StaticIRState(irs), sample_rate=sr, ref_db=-65)
    ev = scene.add_event(core.Event("e0", raw, sr, snr=10.0, scene_start=0.1,
                                    augmentations=[aug.TimeWarpReverse(sr, fps=20, prob=0.5), aug.TimeWarpSilence(sr, fps=10, prob=0.3)]))
    random.seed(5)
    scene.generate()
    chain = ev._last_chain
    assert chain is not None and chain.uploads == 1 and chain.downloads == 0
    a, b = ev.spatial_audio["mic_a"], ev.spatial_audio["mic_b"]
    unit = lambda x: x / np.max(np.abs(x))  # noqa: E731
    for row in list(a) + list(b):
        assert rel_rms(unit(row), unit(a[0])) < 1e-5         # same clip at every capsule of both microphones
    clip = ev.load_audio()                                    # ... and it is the clip load_audio hands out afterwards
    assert ev._last_chain is chain and chain.downloads == 1
    assert rel_rms(unit(a[0]), unit(clip)) < 1e-5
    assert not np.allclose(unit(clip), unit(raw))             # the warp did something
    ev.clear_audio()                                          # dropping the cache draws a new realisation
    assert ev._last_chain is None


def test_a_dropped_scene_frees_its_render_without_the_garbage_collector(golden):
    """The lazy dictionaries of an event hold closures; none of them may hold the event itself (a cycle would keep the render's
    device buffers -- gigabytes per scene at production size -- alive until a collector pass happens to run)."""
    import gc
    import weakref

    scene = build_g8_scene(golden)                      # static + moving + dry-path events, an ambience
    scene.generate()
    ev = next(iter(scene.events.values()))
    held = ev.spatial_audio.device_source("mic000")[0]  # the RenderResult behind event.spatial_audio
    last = list(scene.events.values())[-1]
    assert last._spatial_audio_padded["mic000"].shape == scene.audio["mic000"].shape    # the lazies work through weak references
    del last
    probes = [weakref.ref(held), weakref.ref(ev), weakref.ref(scene)]
    gc.collect()
    gc.disable()
    try:
        del scene, ev, held
        assert [p() for p in probes] == [None, None, None]
    finally:
        gc.enable()
