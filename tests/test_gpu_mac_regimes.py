"""Every dispatch branch of al_spectral_mac (csrc/al_kernels.hip: pick_mac) on the gfx950 build, through the C ABI,
EVERY row against the float64 oracle; each test asserts which instantiation ran (al_spectral_mac_variant).

Covers what the headline bench executes: k_spectral_mac<12,12,2,KSPLIT> at K = 24 / P = 12 (cfg2) and
k_spectral_mac_moving at P = 12 with 32 IRs per event at B = 8192 (cfg3).  Reference: synthesize.py:71-106,184-310.
"""
import pytest

from tests.conftest import set_switch

from tests.conftest import set_switch

from tests import mac_regimes as mr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from audiblelight_amd import engine

    r = engine.Renderer()
    assert r.lib.path.endswith("libaudiblelight_hip.so")
    return r


@pytest.mark.parametrize("log2_block", [10, 11, 12, 13, 14])
@pytest.mark.parametrize("name,code,k_mult,p_mult", mr.STATIC_CASES, ids=[c[0] for c in mr.STATIC_CASES])
def test_static_regimes(gpu, log2_block, name, code, k_mult, p_mult, monkeypatch):
    """The tile kernels (k_spectral_mac + k_block_synthesis): AL_STATIC_MAC=0 keeps their dispatch branches reachable for
    static events (they are the default for more than 12 partitions and for multi-emitter events)."""
    set_switch(monkeypatch, "AL_STATIC_MAC", "0")
    mr.run_static_case(gpu, log2_block, code, k_mult, p_mult, expect_split=(log2_block >= 13))   # 14: csrc/al_quad16.h


@pytest.mark.parametrize("log2_block", [10, 13])
@pytest.mark.parametrize("name,code,k_mult,p_mult,C,E", mr.STATIC_LOOP_CASES, ids=[c[0] for c in mr.STATIC_LOOP_CASES])
def test_static_capsule_loop_kernel(gpu, monkeypatch, log2_block, name, code, k_mult, p_mult, C, E):
    """k_spectral_mac_static / _static_lds / _static_glds (default for one-emitter events with at most 21 partitions): EVERY
    instantiation -- partition counts 1..21 x {one k-tile, two k-tiles, more than 24 blocks} -- at B = 8192 and B = 1024, ragged tiles,
    the capsule-range split of small batches; every row against the oracle, the instantiation asserted."""
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    mr.run_static_case(gpu, log2_block, code, k_mult, p_mult, C=C, E=E)


@pytest.mark.parametrize("log2_block", [10, 13])
@pytest.mark.parametrize("name,code,k_mult,p_mult,C,E", mr.GLDS_CASES, ids=[c[0] for c in mr.GLDS_CASES])
def test_static_capsule_loop_glds_kernel(gpu, monkeypatch, log2_block, name, code, k_mult, p_mult, C, E):
    """k_spectral_mac_static_glds (partition spectra into the LDS ring by LDS-DMA, counted s_waitcnt) for at most 12 partitions,
    where it is an A/B switch (13..24 partitions take it by default: test_static_capsule_loop_kernel): every partition count
    1..12, second k-tile full / ragged / idle, one and several workgroups per (event, bin tile); every row against the oracle."""
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", str(1 << 14))
    mr.run_static_case(gpu, log2_block, code, k_mult, p_mult, C=C, E=E)


@pytest.mark.parametrize("name,code,k_mult,p_mult,C,E", mr.NO_ZERO_BLOCK_CASES, ids=[c[0] for c in mr.NO_ZERO_BLOCK_CASES])
def test_static_capsule_loop_without_zero_block(gpu, monkeypatch, name, code, k_mult, p_mult, C, E):
    """13..16 partitions for a batch without an all-zero spectrum block (hspec_zero_block = -1, e.g. a C host that keeps
    none): the register-staged two-unit kernel k_spectral_mac_static_lds<12,{7,8},2> instead of the LDS-DMA one."""
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", None)
    mr.run_static_case(gpu, 13, code, k_mult, p_mult, C=C, E=E, zero_block=False)


@pytest.mark.parametrize("log2_block", [10, 12, 13])
@pytest.mark.parametrize("p_mult,expect", [(4.3, 612), (11.7, 612), (12.6, 624), (23.9, 624), (24.2, 0)],
                         ids=["P5", "P12", "P13", "P24", "P25_tile_kernel"])
def test_moving_regimes(gpu, monkeypatch, log2_block, p_mult, expect):
    """The sliding-window accumulate over stored IR spectra (k_spectral_mac_moving): the default at every block size."""
    mr.run_moving_case(gpu, log2_block, p_mult, n_irs=10, k_mult=14.2, expect_moving=expect)


@pytest.mark.parametrize("log2_block,split", [(11, "1"), (12, "1"), (13, "0"), (13, "1"), (14, "1")])
def test_transform_layouts(gpu, monkeypatch, log2_block, split):
    """Both spectrum layouts of the FFT kernels: the split layout (csrc/al_split.h: every window as two half-size
    transforms, default at B = 8192) at every block size it is built for, and the one-transform kernels at B = 8192 with
    AL_SPLIT=0 (they stay the default elsewhere and are what every other block size in this file runs).  Static events
    with a ragged last partition / block, moving events and a tiled dry clip, every row against the oracle."""
    import numpy as np

    from audiblelight_amd import plan as planning
    from oracle import synth_oracle as orc

    set_switch(monkeypatch, "AL_SPLIT", split)
    set_switch(monkeypatch, "AL_QUAD16", "0")       # B = 16384 here: two 8192-point transforms (the quad tiles have their own tests below)
    B = 1 << log2_block
    rng = np.random.default_rng(40 + log2_block)
    sr, C, L = 48000, 3, int(2.3 * B) + 7
    specs, clips, irs, col = [], [], [], 0
    for n_audio, n_emit in ((int(4.6 * B) + 3, 1), (7 * B, 6), (B + 5, 0), (B // 2 - 3, 1)):
        a = rng.standard_normal(n_audio).astype(np.float32)
        clips.append(a / np.abs(a).max())
        irs.append((rng.standard_normal((C, n_emit, L)) * np.exp(-np.arange(L) / (L / 5.0))).astype(np.float32))
        specs.append(planning.EventSpec(n_samples=n_audio, n_emitters=n_emit, snr=float(rng.uniform(5, 30)), emitter0=col,
                                        is_moving=n_emit > 1, duration=n_audio / sr))
        col += n_emit
    pl = planning.plan_batch(specs, C, L, sr, log2_block=log2_block)
    batch = gpu.prepare(pl, clips, np.concatenate(irs, axis=1))
    assert mr.is_split(batch) == (split == "1")
    res = batch.run()
    res.check_finite()
    for i, (a, h, sp) in enumerate(zip(clips, irs, specs)):
        want = orc.render_event(a, h.astype(np.float64), sp.snr, is_moving=sp.is_moving, duration=sp.duration, sr=sr)["spatial"]
        mr.check_event_rows(res, i, want)


def test_default_layout_per_block_size(gpu, monkeypatch):
    from audiblelight_amd import plan as planning
    import numpy as np

    set_switch(monkeypatch, "AL_SPLIT", None)
    for lb in (10, 12, 13, 14):
        pl = planning.plan_batch([planning.EventSpec(n_samples=3000, n_emitters=1, snr=5.0)], 2, 500, 48000, log2_block=lb)
        batch = gpu.prepare(pl, [np.zeros(3000, np.float32)], np.zeros((2, 1, 500), np.float32))
        assert mr.is_split(batch) == (lb >= 13)


def test_cfg3_regime_all_rows(gpu):
    """cfg3's own regime: B = 8192, P = 12 partitions (2 s RIR), 32 IRs per event, 7.75 s clips; 2 events x 4 capsules,
    every row against the oracle (the full config differs only in the event / capsule counts)."""
    res = mr.run_moving_case(gpu, 13, 96000 / 8192, n_irs=32, k_mult=372000 / 8192, expect_moving=612, C=4, E=2)
    assert res.plan.n_partitions == 12 and int(res.plan.events["n_blocks"].max()) == 46


def test_cfg4_regime_all_rows(gpu):
    """cfg4's own accumulate at full length: B = 8192, K = 24, P = 6 (4 s clips, 1 s RIRs @ 48 kHz) -> k_spectral_mac_static<12,6,2>;
    3 events x 5 capsules, every row against the oracle."""
    res = mr.run_static_case(gpu, 13, 3120602, 192000 / 8192, 48000 / 8192, C=5, E=3, expect_split=True)
    assert res.plan.n_partitions == 6 and int(res.plan.events["n_blocks"].max()) == 24


def test_cfg2_regime_all_rows(gpu):
    """cfg2's own regime at full length: B = 8192, K = 24, P = 12 (4 s clips, 2 s RIRs @ 48 kHz), 3 events x 5 capsules."""
    res = mr.run_static_case(gpu, 13, 3121202, 192000 / 8192, 96000 / 8192, C=5, E=3, expect_split=True)
    assert res.plan.n_partitions == 12 and int(res.plan.events["n_blocks"].max()) == 24


def test_cfg2_regime_all_rows_tile_kernels(gpu, monkeypatch):
    """The same through k_spectral_mac<12,12,2,KSPLIT> and the one-transform FFT kernels (round 1's path)."""
    set_switch(monkeypatch, "AL_STATIC_MAC", "0")
    set_switch(monkeypatch, "AL_SPLIT", "0")
    mr.run_static_case(gpu, 13, 1121202, 192000 / 8192, 96000 / 8192, C=5, E=3, expect_split=False)


def test_quad16_transforms_all_rows(gpu, monkeypatch):
    """B = 16384 as four 4096-point tiles (csrc/al_quad16.h): k_forward_spectra_quad16 / k_block_synthesis_quad16 around the
    unchanged accumulate.  cfg5's regime (12 partitions in three runs of four, 12 blocks), a short batch with a ragged last
    partition and edge windows, a moving event (the rolled general signal path, sliding-window accumulate), every row against
    the oracle; then the separate IR / signal launches (al_ir_spectra + al_signal_spectra) against the merged one, bit for bit."""
    set_switch(monkeypatch, "AL_QUAD16", None)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", str(4 << 24))       # AL_FLAG_IR_RUN(4): three runs of four partitions (a batch this small gets runs of one)
    res = mr.run_static_case(gpu, 14, 3121201, 192000 / 16384, 192000 / 16384, C=3, E=2, expect_split=True, expect_quad=True)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", str(12 << 24))      # one run of twelve, as the full-size cfg5 batch runs
    mr.run_static_case(gpu, 14, 3121201, 192000 / 16384, 192000 / 16384, C=2, E=1, expect_split=True, expect_quad=True)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", str(5 << 24))       # unequal runs: 5 + 2
    mr.run_static_case(gpu, 14, 3120701, 3.2, 6.01, C=2, E=1, expect_split=True, expect_quad=True)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", None)
    assert res.plan.n_partitions == 12 and int(res.plan.events["n_blocks"].max()) == 12
    mr.run_static_case(gpu, 14, 3120301, 6.5, 2.5, C=3, E=2, expect_split=True, expect_quad=True)
    mr.run_static_case(gpu, 14, 3120701, 3.2, 6.01, C=2, E=1, expect_split=True, expect_quad=True)
    mr.run_moving_case(gpu, 14, 2.3, n_irs=6, k_mult=5.2, expect_moving=612, C=2, E=1)
    mr.run_separate_forward_launches(gpu, 14)
    set_switch(monkeypatch, "AL_QUAD16", "0")       # and the one-transform kernels of round 1 still serve B = 16384
    mr.run_static_case(gpu, 14, 3120301, 6.5, 2.5, C=3, E=2, expect_split=False, expect_quad=False)


def test_quad16_random_batches(gpu, monkeypatch):
    """Seeded random batches (static / moving / zero-emitter events mixed) at B = 16384 through the quad-tile transforms."""
    set_switch(monkeypatch, "AL_QUAD16", None)
    for seed in (3, 11, 29):
        mr.run_random_batch(gpu, seed, log2_block=14)


@pytest.mark.parametrize("seed", range(48))
def test_random_shapes_over_the_whole_dispatch_space(gpu, monkeypatch, seed):
    """Seeded random batches: 1..26 partitions x clips of up to 60 blocks x static / moving / zero-emitter events mixed in one
    batch (B = 1024 for two thirds of the seeds, 8192 for the rest), every row against the oracle: the accumulate kernels
    beside each other as real scenes mix them, not one regime per batch."""
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", None)
    mr.run_random_batch(gpu, seed, log2_block=13 if seed % 3 == 2 else 10)
