"""Every dispatch branch of al_spectral_mac (csrc/al_kernels.hip: pick_mac) on the gfx950 build, through the C ABI,
EVERY row against the float64 oracle; each test asserts which instantiation ran (al_spectral_mac_variant).

Covers what the headline bench executes: k_spectral_mac<12,12,2,KSPLIT> at K = 24 / P = 12 (cfg2) and
k_spectral_mac_moving at P = 12 with 32 IRs per event at B = 8192 (cfg3).  Reference: synthesize.py:71-106,184-310.
"""
import pytest

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
def test_static_regimes(gpu, log2_block, name, code, k_mult, p_mult):
    mr.run_static_case(gpu, log2_block, code, k_mult, p_mult)


@pytest.mark.parametrize("log2_block", [10, 12, 13])
@pytest.mark.parametrize("p_mult,expect", [(4.3, 612), (11.7, 612), (12.6, 624), (23.9, 624), (24.2, 0)],
                         ids=["P5", "P12", "P13", "P24", "P25_tile_kernel"])
def test_moving_regimes(gpu, log2_block, p_mult, expect):
    mr.run_moving_case(gpu, log2_block, p_mult, n_irs=10, k_mult=14.2, expect_moving=expect)


def test_cfg3_regime_all_rows(gpu):
    """cfg3's own regime: B = 8192, P = 12 partitions (2 s RIR), 32 IRs per event, 7.75 s clips; 2 events x 4 capsules,
    every row against the oracle (the full config differs only in the event / capsule counts)."""
    res = mr.run_moving_case(gpu, 13, 96000 / 8192, n_irs=32, k_mult=372000 / 8192, expect_moving=612, C=4, E=2)
    assert res.plan.n_partitions == 12 and int(res.plan.events["n_blocks"].max()) == 46


def test_cfg2_regime_all_rows(gpu):
    """cfg2's own regime at full length: B = 8192, K = 24, P = 12 (4 s clips, 2 s RIRs @ 48 kHz), 3 events x 5 capsules."""
    res = mr.run_static_case(gpu, 13, 1121202, 192000 / 8192, 96000 / 8192, C=5, E=3)
    assert res.plan.n_partitions == 12 and int(res.plan.events["n_blocks"].max()) == 24
