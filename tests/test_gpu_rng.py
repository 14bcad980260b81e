"""Device-side ambience draws on the real MI355X: the scenarios of tests/test_hostemu_rng.py (the reference's statistical
acceptance tests of powerlaw_psd_gaussian / Ambience, tests/test_ambience.py:30-76,107-138) at the reference's own sizes."""
import pytest

from tests import test_hostemu_rng as scenarios

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def gpu_renderer():
    from audiblelight_amd import engine, synthesize as syn

    r = engine.Renderer()
    assert r.lib.path.endswith("libaudiblelight_hip.so")
    syn.set_renderer(r)
    yield r
    syn.set_renderer(None)


test_philox_known_answers = scenarios.test_philox_known_answers
test_normal_fill_is_the_documented_function_of_seed_and_index = scenarios.test_normal_fill_is_the_documented_function_of_seed_and_index
test_var_distribution = scenarios.test_var_distribution
test_small_sample_var = scenarios.test_small_sample_var
test_slope_distribution = scenarios.test_slope_distribution
test_cumulative_scaling = scenarios.test_cumulative_scaling
test_random_state_reproducibility = scenarios.test_random_state_reproducibility
test_normality_of_the_draws = scenarios.test_normality_of_the_draws
test_ambience_cls = scenarios.test_ambience_cls
test_scene_with_device_drawn_ambience_matches_the_oracle_given_the_same_noise = \
    scenarios.test_scene_with_device_drawn_ambience_matches_the_oracle_given_the_same_noise


def test_cfg2_size_white_ambience_is_generated_in_under_a_millisecond_and_never_meets_the_host():
    """32 x 2 880 000 white ambience (cfg2's scene size): draws + statistics + per-channel multipliers are stream-ordered
    device work (no synchronisation between them); timed with HIP events."""
    import torch

    from audiblelight_amd import ambience as amb, synthesize as syn

    r = syn.get_renderer()
    a = amb.Ambience(32, 60.0, alias="a", noise="white", ref_db=-65, sample_rate=48000, rng="device")
    a.noise_and_scales_device(r, (32, 2880000))      # warm-up: allocations
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        a._scaled = None
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        noise, scales = a.noise_and_scales_device(r, (32, 2880000))
        t1.record()
        torch.cuda.synchronize()
        times.append(t0.elapsed_time(t1))
    assert min(times) < 1.0, times
    s = r.mem.download(scales)[:32]
    x = r.mem.download(noise)[: 32 * 2880000].reshape(32, -1)
    peak = np.abs(x).max(axis=1)
    mean_norm = np.mean(np.abs(x) / peak[:, None])
    np.testing.assert_allclose(s, 10 ** (-65 / 20) / mean_norm / peak, rtol=1e-5)
    assert abs(x.std() - 1) < 1e-3 and 4.5 < peak.min() and peak.max() < 6.8


import numpy as np  # noqa: E402
