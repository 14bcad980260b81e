"""Scenarios that pin every dispatch branch of al_spectral_mac (csrc/al_kernels.hip: pick_mac) against the float64
oracle, EVERY row compared.  Shared by tests/test_hostemu_regimes.py (host emulation, small blocks) and
tests/test_gpu_mac_regimes.py (gfx950 build, every block size incl. the cfg2 / cfg3 regimes).

Reference semantics: audiblelight/synthesize.py:71-106 (static) and :184-310 (moving).
"""
import ctypes as ct

import numpy as np

from audiblelight_amd import plan as planning
from oracle import synth_oracle as orc
from tests.conftest import assert_parity, rel_rms

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

# k_spectral_mac_static<12,P,{1,2}> / k_spectral_mac_static_lds<12,P> / <12,ceil(P/2),2> / <12,6,3>: EVERY partition count 1..18 in each of
# the three clip-length regimes that pick a different instantiation (or, for 13..18 partitions, a different grid of the same one):
# (name, expected code 3120000 + 100*P + {1: one k-tile per workgroup, 2: two, 3: partition spectra staged through LDS},
#  K multiple, P multiple, capsules, events); one event => the capsule loop is split into ranges (small batch)
_CLIP_REGIMES = (("one_ktile", 9.3, 1), ("two_ktiles", 20.6, 2), ("beyond_24_blocks", 26.3, 3))
STATIC_LOOP_CASES = [
    (f"P{P}_{name}", 3120000 + 100 * P + (4 if P > 12 else digit), k_mult + 0.01 * P, P - 0.37, 2 + P % 2, 1 + (P % 3 == 0))
    for P in range(1, 22) for name, k_mult, digit in _CLIP_REGIMES]
STATIC_LOOP_CASES += [   # hand-picked edges
    ("cfg2_shape_pair_full", 3121202, 23.44, 11.72, 5, 2),     # K = 24 (two k-tiles in one workgroup), P = 12
    ("pair_ragged_ktile_masked", 3120902, 17.3, 8.6, 3, 2),    # K = 18 (second k-tile half empty), P = 9
    ("three_ktiles_idle_half", 3121203, 26.3, 12.0, 2, 1),     # K = 27: 3 k-tiles in 2 workgroups, the last has an idle half
    ("lds_ring_two_partitions", 3120203, 50.4, 1.7, 2, 1),     # K = 51 (3 workgroups), P = 2; one event: capsule ranges split
    ("two_units_16_long", 3121604, 30.2, 15.4, 2, 1),          # P = 16, K = 31: two workgroups per (event, bin tile)
    ("two_units_one_ktile", 3121404, 7.5, 13.2, 2, 2),         # P = 14, K = 8: the second half of the workgroup idles
    ("cfg5_shape_tile_kernel", 1121202, 23.44, 23.44, 3, 2),   # K = 24, P = 24: cfg5's tile counts (two full partition tiles)
]
# k_spectral_mac_static_glds<12,P> for at most 12 partitions (AL_EXTRA_FLAGS bit 14: an A/B switch, the default there is the
# register / register-staged kernel): P = 1..12, clips of 13..24 and of more than 24 blocks
GLDS_CASES = [(f"glds_P{P}_{name}", 3120000 + 100 * P + 4, k_mult + 0.01 * P, P - 0.37, 2 + P % 2, 1 + (P % 3 == 0))
              for P in range(1, 13) for name, k_mult, digit in _CLIP_REGIMES[1:]]
GLDS_CASES += [("glds_cfg2_shape", 3121204, 23.44, 11.72, 5, 2), ("glds_ragged_second_tile", 3120904, 17.3, 8.6, 3, 2),
               ("glds_three_ktiles_idle_half", 3121204, 26.3, 12.0, 2, 1)]
# k_spectral_mac_static_lds<12,{7,8},2>: 13..16 partitions for a caller that gave no all-zero block (hspec_zero_block = -1)
NO_ZERO_BLOCK_CASES = [(f"no_zero_block_P{P}", 3120000 + 100 * P + 3, 20.6, P - 0.37, 2, 1) for P in (13, 14, 15, 16)]
MOVING_CODES = [612, 624]       # asserted by test_moving_regimes / test_cfg3_regime_all_rows
# codes the GPU tests assert beyond the tables above: cfg4's <12,6,2>, cfg5's tile kernel with two full partition tiles
EXTRA_STATIC_CODES = [3120602, 1121202]


def codes_of_kernel_symbol(sym: str):
    """Demangled kernel name (``nm -C``) -> the al_spectral_mac_variant codes under which it runs, as (kind, code) pairs."""
    import re

    m = re.search(r"k_spectral_mac(_static_lds|_static_glds|_static|_moving)?<([0-9, a-z]+)>", sym)
    if not m:
        return []
    kind, args = m.group(1) or "", [a.strip() for a in m.group(2).split(",")]
    if kind == "":
        kt, pt, vb = int(args[0]), int(args[1]), int(args[2])
        return [("static", (1000000 if args[3] == "true" else 0) + 10000 * kt + 100 * pt + vb)]
    if kind == "_moving":
        return [("moving", 100 * int(args[0]) + int(args[1]))]
    pt, last = int(args[1]), int(args[2])
    if kind == "_static_glds":
        units = int(args[2])
        return [("static", 3120000 + 100 * p + 4) for p in ([pt] if units == 1 else range(units * (pt - 1) + 1, units * pt + 1))
                if units < 3 or p > 16]
    if kind == "_static":
        return [("static", 3120000 + 100 * pt + last)]
    if last == 1:
        return [("static", 3120000 + 100 * pt + 3)]
    if last == 2:
        return [("static", 3120000 + 100 * p + 3) for p in (2 * pt - 1, 2 * pt)]   # two units per capsule: P = 2*PT - 1 and 2*PT
    raise AssertionError(f"unexpected instantiation {sym}")


def asserted_codes():
    """Every (kind, code) some -m gpu test asserts through al_spectral_mac_variant."""
    out = {("static", c[1]) for c in STATIC_CASES} | {("static", c[1]) for c in STATIC_LOOP_CASES} | {("static", c[1]) for c in GLDS_CASES}
    out |= {("static", c[1]) for c in NO_ZERO_BLOCK_CASES}
    out |= {("static", c) for c in EXTRA_STATIC_CODES} | {("moving", c) for c in MOVING_CODES}
    return out


def mac_codes(renderer, batch, chunk=0):
    s, m = ct.c_int32(-1), ct.c_int32(-1)
    renderer.lib.call("al_spectral_mac_variant", ct.byref(batch.descs[chunk]), ct.byref(s), ct.byref(m))
    return s.value, m.value


def check_event_rows(res, i, want, tol=TOL):
    """Every (capsule) row of event i against the oracle: per-row relative RMS and the max-abs bound."""
    got = res.spatial_audio(i)
    assert got.shape == want.shape
    for c in range(want.shape[0]):
        assert_parity(got[c], want[c], tol, what=(i, c))
    assert_parity(got, want, tol, what=i)


def is_split(batch, chunk=0):
    from audiblelight_amd import _hip

    return bool(batch.descs[chunk].flags & _hip.FLAG_SPLIT_SPECTRA)


def run_static_case(renderer, log2_block, code, k_mult, p_mult, C=3, E=2, seed=0, expect_split=None, zero_block=True,
                    expect_quad=None):
    """A batch of E static events of about k_mult blocks against IRs of about p_mult partitions: the accumulate instantiation the
    library reports must be ``code``, the layout flags what the caller expects, every row what the oracle renders."""
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
    if not zero_block:       # a C host that does not provide the all-zero spectrum block
        for desc in batch.descs:
            desc.hspec_zero_block = -1
    got_code, moving = mac_codes(renderer, batch)
    assert got_code == code and moving == 0, (got_code, moving)
    if expect_split is not None:
        assert is_split(batch) == expect_split
    if expect_quad is not None:
        from audiblelight_amd import _hip
        assert bool(batch.descs[0].flags & _hip.FLAG_QUAD_SPECTRA) == expect_quad
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
        # al_batch.emitter_parts: partitions that only reach blocks past the clip's end are neither transformed nor read.  The
        # spectra workspace is poisoned first, so a read of a block that was not written would turn the event into NaNs.
        parts = pl.emitter_parts()
        assert parts is not None and 0 <= parts.min() < pl.n_partitions and parts.max() <= pl.n_partitions
        assert batch.descs[0].emitter_parts, "the planner's table did not reach the descriptor"
        n_real = pl.hspec_blocks * B * 2
        batch.bufs["hspec"][:n_real] = float("nan")
    else:
        assert pl.emitter_parts() is None and not batch.descs[0].emitter_parts
    res = batch.run()
    res.check_finite()
    if expect_moving:   # ... and the transform really left those blocks alone: exactly the trimmed ones still hold the poison
        h = np.asarray(renderer.mem.download(batch.bufs["hspec"]))[:n_real].reshape(pl.n_emitters, C, pl.n_partitions, 2 * B)
        untouched = np.isnan(h).all(axis=3)
        expected = np.arange(pl.n_partitions)[None, None, :] >= parts[:, None, None]
        assert np.array_equal(untouched, np.broadcast_to(expected, untouched.shape)) and not np.isnan(h[~untouched]).any()
    for e in range(E):
        want = orc.render_event(clips[e], irs[e].astype(np.float64), specs[e].snr, is_moving=True,
                                duration=specs[e].duration, sr=sr)["spatial"]
        check_event_rows(res, e, want)
    return res


def run_separate_forward_launches(renderer, log2_block, seed=0):
    """al_ir_spectra + al_signal_spectra (two launches, what a host that streams IRs in chunks calls) against al_forward_spectra
    (one launch) on a mixed static / moving batch: the event audio must come out bit for bit the same."""
    rng = np.random.default_rng(31000 + seed)
    B, sr, C = 1 << log2_block, 48000, 2
    Lir, La = int(2.4 * B), int(4.3 * B)
    clips = [rng.standard_normal(La - 7 * e).astype(np.float32) for e in range(2)]
    irs = (rng.standard_normal((C, 1 + 3, Lir)) * np.exp(-np.arange(Lir) / (Lir / 5.0))).astype(np.float32)
    specs = [planning.EventSpec(n_samples=len(clips[0]), n_emitters=1, snr=10.0, emitter0=0),
             planning.EventSpec(n_samples=len(clips[1]), n_emitters=3, snr=7.0, emitter0=1, is_moving=True, duration=len(clips[1]) / sr)]
    pl = planning.plan_batch(specs, C, Lir, sr, log2_block=log2_block)
    batch = renderer.prepare(pl, clips, irs)
    res = batch.run()
    res.check_finite()
    merged = np.array(renderer.mem.download(res.spatial))
    assert np.nanmax(np.abs(merged)) > 0
    for name in ("spatial", "hspec", "xspec", "yspec"):          # nothing of the first pass may survive into the second
        batch.bufs[name][: -2 << log2_block if name in ("hspec", "xspec") else None] = float("nan")
    stages = ["al_ir_spectra", "al_signal_spectra"] + [s for s in batch.stage_names() if s != "al_forward_spectra"]
    separate = np.array(renderer.mem.download(batch.run(stages=stages).spatial))
    rows = np.concatenate([np.arange(int(ev["out_off"]), int(ev["out_off"]) + C * int(ev["len"])) for ev in pl.events])
    assert np.isfinite(merged[rows]).all() and np.array_equal(merged[rows], separate[rows])


def run_random_batch(renderer, seed, log2_block=10):
    """Seeded random batch over the WHOLE shape space of the accumulate dispatch: 1..26 partitions, clips of 1..60 blocks
    (ragged, several per batch), 1..5 capsules, static events mixed with moving and zero-emitter ones; every row against
    the oracle.  Whatever instantiation the library picks for it is what gets checked."""
    rng = np.random.default_rng(5000 + seed)
    B = 1 << log2_block
    sr = 16000
    C = int(rng.integers(1, 6))
    Lir = int(rng.integers(1, 26 * B))
    specs, clips, irs, col = [], [], [], 0
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.choice(["static", "static", "static", "moving", "dry"])
        n_audio = int(rng.integers(1, 60 * B if rng.random() < 0.3 else 26 * B))
        if kind == "moving":
            n_audio = max(n_audio, 700)
        n_emit = {"static": 1, "dry": 0, "moving": int(rng.integers(2, 6))}[kind]
        a = rng.standard_normal(n_audio).astype(np.float32)
        clips.append(a / np.abs(a).max())
        irs.append((rng.standard_normal((C, n_emit, Lir)) * np.exp(-np.arange(Lir) / max(Lir / 5.0, 1.0))).astype(np.float32))
        specs.append(planning.EventSpec(n_samples=n_audio, n_emitters=n_emit, snr=float(rng.uniform(5, 30)), emitter0=col,
                                        is_moving=n_emit > 1, duration=n_audio / sr))
        col += n_emit
    pl = planning.plan_batch(specs, C, Lir, sr, log2_block=log2_block)
    batch = renderer.prepare(pl, clips, np.concatenate(irs, axis=1))
    codes = mac_codes(renderer, batch)
    res = batch.run()
    res.check_finite()
    for i, (a, h, sp) in enumerate(zip(clips, irs, specs)):
        want = orc.render_event(a, h.astype(np.float64), sp.snr, is_moving=sp.is_moving, duration=sp.duration, sr=sr)["spatial"]
        check_event_rows(res, i, want)
    return codes, pl.n_partitions, int(pl.events["n_blocks"].max())
