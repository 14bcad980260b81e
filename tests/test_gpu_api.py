"""The reference-named API on the real MI355X: the same scenarios as tests/test_hostemu_api.py, run
through the gfx950 library (audiblelight_amd/csrc/libaudiblelight_hip.so) instead of the host emulation."""
import pytest

from tests.conftest import set_switch

from tests import test_hostemu_api as scenarios

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def gpu_renderer():
    from audiblelight_amd import engine, synthesize as syn

    r = engine.Renderer()
    assert r.lib.path.endswith("libaudiblelight_hip.so")
    syn.set_renderer(r)
    yield r
    syn.set_renderer(None)


test_scene_generate_matches_reference = scenarios.test_scene_generate_matches_reference
test_render_cache_and_host_arrays = scenarios.test_render_cache_and_host_arrays
test_render_event_audio_and_errors = scenarios.test_render_event_audio_and_errors
test_validate_scene_messages = scenarios.test_validate_scene_messages
test_outputs_carry_the_reference_dtype = scenarios.test_outputs_carry_the_reference_dtype
test_standalone_convolutions = scenarios.test_standalone_convolutions
test_pointwise_fx_match_definitions = scenarios.test_pointwise_fx_match_definitions
test_timewarp_matches_reference_semantics = scenarios.test_timewarp_matches_reference_semantics
test_event_fx_chain_and_dict_roundtrip = scenarios.test_event_fx_chain_and_dict_roundtrip
test_powerlaw_noise_matches_reference = scenarios.test_powerlaw_noise_matches_reference
test_powerlaw_misc_and_ambience_class = scenarios.test_powerlaw_misc_and_ambience_class
test_scene_with_device_ambience = scenarios.test_scene_with_device_ambience
test_two_microphones_with_different_capsule_counts = scenarios.test_two_microphones_with_different_capsule_counts
test_scene_json_round_trip = scenarios.test_scene_json_round_trip
test_scene_generate_argument_list = scenarios.test_scene_generate_argument_list
test_stft_helpers_match_reference = scenarios.test_stft_helpers_match_reference
test_moving_events_under_other_stft_geometries = scenarios.test_moving_events_under_other_stft_geometries
test_degenerate_events_render_like_the_reference = scenarios.test_degenerate_events_render_like_the_reference
test_general_stft_path_limits = scenarios.test_general_stft_path_limits
test_fx_chain_stays_on_device_and_scalars_fold = scenarios.test_fx_chain_stays_on_device_and_scalars_fold
test_reference_format_scene_json_renders_like_the_reference = scenarios.test_reference_format_scene_json_renders_like_the_reference
test_ir_ingest_ragged_packing_and_resampling = scenarios.test_ir_ingest_ragged_packing_and_resampling


def test_large_noise_lengths_statistics():
    """Full-size ambience (60 s @ 48 kHz = 2,880,000 samples, smooth length) and an awkward prime-ish
    length through Bluestein: unit variance and the requested spectral slope (reference tests/test_ambience.py:30-58)."""
    import numpy as np

    from audiblelight_amd import ambience as amb
    from oracle import synth_oracle as orc

    x = amb.powerlaw_psd_gaussian(0, (2, 2_880_000), seed=1)
    assert abs(x.std() - 1.0) < 0.01 and abs(x.mean()) < 0.01
    small = amb.powerlaw_psd_gaussian(1, (2, 100_003), seed=3)      # 100003 is prime -> Bluestein
    ref = orc.powerlaw_noise(1, (2, 100_003), seed=3)
    assert np.sqrt(np.mean((small - ref) ** 2)) / ref.std() < 1e-5
test_encode_frames_every_store_path = scenarios.test_encode_frames_every_store_path
test_event_from_wav_file_resamples_on_the_device = scenarios.test_event_from_wav_file_resamples_on_the_device
test_fx_match_the_reference_classes_outputs = scenarios.test_fx_match_the_reference_classes_outputs
test_ambience_file_mode_matches_the_reference = scenarios.test_ambience_file_mode_matches_the_reference
test_big_batches_chunk_themselves = scenarios.test_big_batches_chunk_themselves
test_dcase_metadata_matches_the_reference_function = scenarios.test_dcase_metadata_matches_the_reference_function
test_one_fx_realisation_per_event_across_microphones = scenarios.test_one_fx_realisation_per_event_across_microphones


def test_background_ir_upload_equals_inline_upload_on_any_stream(gpu_renderer, monkeypatch):
    """render_audio_for_all_scene_events starts the IR tensors on their way to HBM from a helper thread while it plans
    (Renderer.upload_irs_beside: float32 with 4-float rows as one pageable copy; float64 and ragged rows stay inline): same
    bits as the inline upload, on the default stream and on another one."""
    import numpy as np
    import torch

    from audiblelight_amd import core, engine

    rng = np.random.default_rng(5)
    sr, n_caps, n_ir = 16000, 4, 4000
    irs = {"a": rng.standard_normal((n_caps, 3, n_ir)).astype(np.float32) * np.exp(-np.arange(n_ir) / 600.0).astype(np.float32),
           "b": rng.standard_normal((2, 3, n_ir)) * np.exp(-np.arange(n_ir) / 900.0),            # float64, as get_irs() returns
           "c": rng.standard_normal((2, 3, n_ir - 2)).astype(np.float32)}                         # rows to re-pitch: inline
    clips = [rng.standard_normal(n).astype(np.float32) for n in (9000, 12000, 5000)]

    def render():
        sc = core.Scene(2.0, core.StaticIRState(irs), sample_rate=sr, ref_db=-50)
        for i, clip in enumerate(clips):
            sc.add_event(core.Event(f"e{i}", clip, sr, snr=5 + i, scene_start=0.1 * i))
        out = sc.generate()
        return {k: np.array(v) for k, v in out.items()}

    set_switch(monkeypatch, "AL_BESIDE_MIN_BYTES", "0")       # default: tensors of 8 MiB and more
    calls = []
    real = engine.Renderer.upload_irs_beside

    def spy(self, arr):
        started = real(self, arr)
        calls.append((arr.dtype.name, started is not None))
        return started

    monkeypatch.setattr(engine.Renderer, "upload_irs_beside", spy)
    beside = render()
    assert calls == [("float32", True), ("float64", False), ("float32", False)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        on_side = render()
    side.synchronize()
    monkeypatch.setattr(engine.Renderer, "upload_irs_beside", lambda self, arr: None)
    inline = render()
    for k in irs:
        assert np.array_equal(beside[k], inline[k]) and np.array_equal(on_side[k], inline[k])


def test_free_bytes_reads_the_allocator_counters(gpu_renderer):
    """TorchMemory.free_bytes (the budget behind Renderer.auto_chunk_events) = free on the device + cached by torch's allocator."""
    import torch

    keep = torch.empty(1 << 20, device="cuda")
    del keep                                      # now a cached, unallocated block
    free, _ = torch.cuda.mem_get_info()
    want = free + torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
    assert gpu_renderer.mem.free_bytes() == want and torch.cuda.memory_reserved() > torch.cuda.memory_allocated()


def test_general_stft_path_beyond_one_grid_of_frames():
    """A clip of 70 001 STFT frames (hop 2: more than the 65 535 rows one launch's grid holds; hop 16 at 44.1 kHz gets there after
    23 s) through render_event_audio's literal STFT chain -- al_stft, al_tv_stft_mac and al_istft_ola all walk their frame axes in
    groups of launches -- against the oracle's literal restatement (synthesize.py:184-310), every sample.  (GPU only: the host
    emulation needs 90 s for it.)"""
    import types

    import numpy as np

    from audiblelight_amd import synthesize as syn
    from oracle import synth_oracle as orc
    from tests.conftest import assert_parity

    rng = np.random.default_rng(1)
    n = 140_001
    a = rng.standard_normal(n).astype(np.float32)
    a /= np.abs(a).max()
    h = rng.standard_normal((2, 2, 20)).astype(np.float32)
    ev = types.SimpleNamespace(alias="m", snr=7.0, sample_rate=8000, is_moving=True, duration=n / 8000, spatial_audio={},
                               _spatial_audio_dry={}, ref_ir_channel=None, direct_path_time_ms=None,
                               load_audio=lambda ignore_cache=False, normalize=True: a, __len__=lambda: 2)
    syn.render_event_audio(ev, h, "m", ref_db=-60, fft_size=6, win_size=3, hop_size=2)
    want = orc.render_event(a, h.astype(np.float64), 7.0, ref_db=-60, is_moving=True, duration=n / 8000, sr=8000, nfft=6, win=3, hop=2)
    assert_parity(ev.spatial_audio["m"], want["spatial"], 1e-4, what="70 001 frames")
