"""Scenarios that pin every dispatch branch of al_spectral_mac (csrc/al_kernels.hip: pick_mac) against the float64
oracle, EVERY row compared.  Shared by tests/test_hostemu_regimes.py (host emulation, small blocks) and
tests/test_gpu_mac_regimes.py (gfx950 build, every block size incl. the cfg2 / cfg3 regimes).

Reference semantics: audiblelight/synthesize.py:71-106 (static) and :184-310 (moving).
"""
import ctypes as ct

import numpy as np

from audiblelight_amd import plan as planning
from oracle import synth_oracle as orc
from tests.conftest import rel_rms

TOL = 1e-4

# (name, expected static code, K blocks as a multiple of B, P partitions as a multiple of B); lengths get a ragged tail
STATIC_CASES = [
    ("ksplit_12_12_2", 1121202, 10.0006, 5.002),     # K = 11 > 8, P = 6 > 4: the headline (cfg2) instantiation
    ("ksplit_cfg2_shape", 1121202, 23.44, 11.72),    # K = 24, P = 12: cfg2's own tile counts (2 k-tiles, 1 p-tile)
    ("ksplit_3_ktiles_2_ptiles", 1121202, 26.3, 13.1),  # K = 27 (3 k-tiles), P = 14 (second p-tile partly empty)
    ("tile_8_12_1", 81201, 5.001, 6.0004),           # K = 6 <= 8, P = 7 > 4
    ("tile_24_4_1", 240401, 20.0008, 3.0001),        # K = 21 > 8, P = 4
    ("tile_24_4_1_two_ktiles", 240401, 30.2, 2.5),   # K = 31: the k0 loop runs twice
    ("tile_8_4_1", 80401, 3.0, 2.0),                 # exact multiples: K = 3, P = 2
]


def mac_codes(renderer, batch, chunk=0):
    s, m = ct.c_int32(-1), ct.c_int32(-1)
    renderer.lib.call("al_spectral_mac_variant", ct.byref(batch.descs[chunk]), ct.byref(s), ct.byref(m))
    return s.value, m.value


def check_event_rows(res, i, want, tol=TOL):
    """Every (capsule) row of event i against the oracle: per-row relative RMS and the max-abs bound."""
    got = res.spatial_audio(i)
    assert got.shape == want.shape
    for c in range(want.shape[0]):
        assert rel_rms(got[c], want[c]) <= tol, (i, c)
    assert np.max(np.abs(got - want)) <= 10 * tol * np.max(np.abs(want))


def is_split(batch, chunk=0):
    from audiblelight_amd import _hip

    return bool(batch.descs[chunk].flags & _hip.FLAG_SPLIT_SPECTRA)


def is_fused(batch, chunk=0):
    from audiblelight_amd import _hip

    return bool(batch.descs[chunk].flags & _hip.FLAG_FUSED_STATIC)


def run_static_case(renderer, log2_block, code, k_mult, p_mult, C=3, E=2, seed=0, expect_fused=None, expect_split=None):
    """``expect_fused``: None = whatever the library picks (B = 8192: al_mac_synthesis, csrc/al_fused.h), True / False =
    assert it (callers force the unfused kernels with AL_FUSED=0 so that their dispatch branches stay pinned too)."""
    B = 1 << log2_block
    rng = np.random.default_rng(100 * log2_block + seed)
    La, Lir = int(round(k_mult * B)), int(round(p_mult * B))
    clips, irs, specs = [], [], []
    for e in range(E):
        n = La - 13 * e                      # ragged clip lengths inside one batch
        a = rng.standard_normal(n).astype(np.float32)
        clips.append(a / np.abs(a).max())
        irs.append((rng.standard_normal((C, 1, Lir)) * np.exp(-np.arange(Lir) / (Lir / 5.0))).astype(np.float32))
        specs.append(planning.EventSpec(n_samples=n, n_emitters=1, snr=float(rng.uniform(5, 30)), emitter0=e))
    mic_ir = np.concatenate(irs, axis=1)
    pl = planning.plan_batch(specs, C, Lir, 48000, log2_block=log2_block)
    batch = renderer.prepare(pl, clips, mic_ir)
    got_code, moving = mac_codes(renderer, batch)
    assert got_code == code and moving == 0, (got_code, moving)
    if expect_split is not None:
        assert is_split(batch) == expect_split
    if expect_fused is not None:
        assert is_fused(batch) == expect_fused
        assert ("al_mac_synthesis" in batch.stage_names()) == expect_fused
        assert ("al_spectral_mac" in batch.stage_names()) == (not expect_fused)
    res = batch.run()
    res.check_finite()
    for e in range(E):
        want = orc.render_event(clips[e], irs[e].astype(np.float64), specs[e].snr, sr=48000)["spatial"]
        check_event_rows(res, e, want)
    return res


def run_moving_case(renderer, log2_block, p_mult, n_irs, k_mult, expect_moving, C=2, E=2, seed=0, sr=48000):
    """Moving events whose cross-fade windows are short against the block (sliding-window kernel) or, with
    expect_moving == 0, too many partitions for it (tile kernel summing over streams)."""
    B = 1 << log2_block
    rng = np.random.default_rng(7000 + 10 * log2_block + seed)
    La, Lir = int(round(k_mult * B)), int(round(p_mult * B))
    clips, irs, specs, col = [], [], [], 0
    for e in range(E):
        n = La - 301 * e
        a = rng.standard_normal(n).astype(np.float32)
        clips.append(a / np.abs(a).max())
        irs.append((rng.standard_normal((C, n_irs, Lir)) * np.exp(-np.arange(Lir) / (Lir / 5.0))).astype(np.float32))
        specs.append(planning.EventSpec(n_samples=n, n_emitters=n_irs, snr=float(rng.uniform(5, 30)), emitter0=col,
                                        is_moving=True, duration=n / sr))
        col += n_irs
    mic_ir = np.concatenate(irs, axis=1)
    pl = planning.plan_batch(specs, C, Lir, sr, log2_block=log2_block)
    batch = renderer.prepare(pl, clips, mic_ir)
    _, moving = mac_codes(renderer, batch)
    assert moving == expect_moving, moving
    if expect_moving:
        assert all(int(r) == 1 for r in pl.events["reserved"]), "planner did not flag the events for the sliding window"
    res = batch.run()
    res.check_finite()
    for e in range(E):
        want = orc.render_event(clips[e], irs[e].astype(np.float64), specs[e].snr, is_moving=True,
                                duration=specs[e].duration, sr=sr)["spatial"]
        check_event_rows(res, e, want)
    return res
