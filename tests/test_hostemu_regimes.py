"""Every dispatch branch of al_spectral_mac on the host-emulated kernels (B = 1024): index arithmetic of the k-tile /
p-tile loops, KSPLIT workgroup mapping and the sliding-window kernel, every row against the oracle.  The gfx950
build repeats these at every block size in tests/test_gpu_mac_regimes.py."""
import pytest

from tests.conftest import set_switch

from audiblelight_amd import _hip, engine
from tests import hostemu, mac_regimes as mr


@pytest.fixture(scope="module")
def emu():
    return engine.Renderer(lib=_hip.Library(hostemu.build()), memory=hostemu.NumpyMemory())


@pytest.mark.parametrize("name,code,k_mult,p_mult", mr.STATIC_CASES, ids=[c[0] for c in mr.STATIC_CASES])
def test_emu_static_regimes(emu, name, code, k_mult, p_mult, monkeypatch):
    set_switch(monkeypatch, "AL_STATIC_MAC", "0")   # the tile kernels
    mr.run_static_case(emu, 10, code, k_mult, p_mult, C=2)


@pytest.mark.parametrize("code,k_mult,p_mult,C,E", [(3120902, 17.3, 8.6, 3, 1), (3120601, 10.0006, 5.002, 2, 2), (3120301, 6.5, 2.5, 3, 1),
                                                   (3120503, 25.7, 4.3, 2, 1), (3121304, 14.3, 12.6, 2, 1), (3121704, 13.2, 16.6, 2, 1), (3122004, 13.2, 19.4, 1, 1)])
def test_emu_static_capsule_loop_kernel(emu, code, k_mult, p_mult, C, E, monkeypatch):
    """k_spectral_mac_static under emulation: paired k-tiles with a half-empty second tile and masked partitions, the
    6-partition instantiations, the capsule-range split."""
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    mr.run_static_case(emu, 10, code, k_mult, p_mult, C=C, E=E)


@pytest.mark.parametrize("name,code,k_mult,p_mult,C,E", mr.STATIC_LOOP_CASES, ids=[c[0] for c in mr.STATIC_LOOP_CASES])
def test_emu_every_capsule_loop_instantiation(emu, monkeypatch, name, code, k_mult, p_mult, C, E):
    """The case table the GPU suite runs at B = 8192 and 1024 (partition counts 1..21 x three clip-length regimes + edges), here
    at B = 1024 on the host-emulated kernels: the index arithmetic of every capsule-loop instantiation is checked on CPU too."""
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", None)
    mr.run_static_case(emu, 10, code, k_mult, p_mult, C=min(C, 2), E=1)


@pytest.mark.parametrize("name,code,k_mult,p_mult,C,E", mr.GLDS_CASES[::3] + mr.NO_ZERO_BLOCK_CASES[::3],
                         ids=[c[0] for c in mr.GLDS_CASES[::3] + mr.NO_ZERO_BLOCK_CASES[::3]])
def test_emu_glds_switch_and_no_zero_block(emu, monkeypatch, name, code, k_mult, p_mult, C, E):
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    if code % 10 == 4:
        set_switch(monkeypatch, "AL_EXTRA_FLAGS", str(1 << 14))
        mr.run_static_case(emu, 10, code, k_mult, p_mult, C=min(C, 2), E=1)
    else:
        set_switch(monkeypatch, "AL_EXTRA_FLAGS", None)
        mr.run_static_case(emu, 10, code, k_mult, p_mult, C=min(C, 2), E=1, zero_block=False)


def test_emu_static_glds_kernel(emu, monkeypatch):
    """k_spectral_mac_static_glds under emulation (the LDS-DMA pieces as plain copies: ring indexing, piece -> row mapping,
    the repeated last piece where PT * 4 is not a multiple of 8, ragged second k-tile)."""
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", str(1 << 14))
    mr.run_static_case(emu, 10, 3120904, 17.3, 8.6, C=3, E=1)
    mr.run_static_case(emu, 10, 3120304, 26.3, 2.5, C=2, E=1)


@pytest.mark.parametrize("p_mult,expect", [(4.3, 612), (12.6, 624), (24.2, 0)])
def test_emu_moving_regimes(emu, p_mult, expect):
    mr.run_moving_case(emu, 10, p_mult, n_irs=10, k_mult=14.2, expect_moving=expect, C=2, E=1)


def test_emu_quad16_transforms(emu, monkeypatch):
    """csrc/al_quad16.h under emulation (B = 16384 as four 4096-point tiles): a run of three IR partitions with a ragged last one
    (the prefetch hand-over between partitions), interior and edge signal windows, the rolled general signal path with cross-fade
    envelopes (moving event), the inverse that assembles a block from the four tiles; every row against the oracle.  And with
    AL_QUAD16=0 the one-transform kernels of round 1 still serve B = 16384."""
    set_switch(monkeypatch, "AL_QUAD16", None)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", str(3 << 24))       # AL_FLAG_IR_RUN(3): one run of three partitions per IR row (a batch this small gets runs of one)
    mr.run_static_case(emu, 14, 3120301, 3.5, 2.5, C=2, E=2, expect_split=True, expect_quad=True)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", str(2 << 24))       # unequal runs: 2 + 1
    mr.run_static_case(emu, 14, 3120301, 3.5, 2.5, C=1, E=1, expect_split=True, expect_quad=True)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", None)
    mr.run_moving_case(emu, 14, 2.3, n_irs=6, k_mult=5.2, expect_moving=612, C=2, E=1)
    mr.run_separate_forward_launches(emu, 14)
    set_switch(monkeypatch, "AL_QUAD16", "0")
    mr.run_static_case(emu, 14, 3120201, 2.2, 1.5, C=1, E=1, expect_split=False, expect_quad=False)


def test_emu_split_layout_transforms(emu, monkeypatch):
    """csrc/al_split.h under emulation (B = 2048, the smallest it is built for): even / odd half spectra, the LDS layout
    change of the difference signal, the combine of the two half-size inverses; static (two k-tiles, ragged partitions)
    and moving events, every row against the oracle."""
    set_switch(monkeypatch, "AL_SPLIT", "1")
    mr.run_static_case(emu, 11, 3120601, 10.0006, 5.002, C=2, E=1, expect_split=True)
    mr.run_moving_case(emu, 11, 4.3, n_irs=10, k_mult=14.2, expect_moving=612, C=1, E=1)


@pytest.mark.parametrize("seed", range(10))
def test_emu_random_shapes_over_the_whole_dispatch_space(emu, monkeypatch, seed):
    set_switch(monkeypatch, "AL_STATIC_MAC", None)
    set_switch(monkeypatch, "AL_EXTRA_FLAGS", None)
    mr.run_random_batch(emu, seed)
