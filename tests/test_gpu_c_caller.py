"""The C ABI from a plain C host (tests/c_caller/render_static.c, gcc, no Python / torch / C++ in the process): static
events rendered with al_render_batch + al_mixdown, every row compared with the float64 oracle.  The same comparison as
tests/test_gpu_parity.py makes through ctypes -- here to pin that include/audiblelight_hip.h alone is enough to bind
the library (struct layouts, buffer sizes, call order)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import synth_oracle as orc
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
TOL = 1e-4
EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c_caller", "_build", "render_static")


@pytest.mark.parametrize("C,E,La,Lir,log2_block", [(3, 2, 9001, 2500, 10), (4, 3, 50_000, 30_001, 13)])
def test_c_host_renders_what_the_oracle_renders(tmp_path, C, E, La, Lir, log2_block):
    if not os.path.exists(EXE):
        import __graft_entry__

        __graft_entry__.build_c_caller()
    rng = np.random.default_rng(C * 1000 + E)
    sr, ref_db = 48000, -65.0
    snr = rng.uniform(5, 30, E).astype(np.float32)
    clips = rng.standard_normal((E, La)).astype(np.float32)
    clips /= np.abs(clips).max(axis=1, keepdims=True)
    irs = (rng.standard_normal((C, E, Lir)) * np.exp(-np.arange(Lir) / (Lir / 6.0))).astype(np.float32)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(src, "wb") as f:
        np.array([C, E, La, Lir, log2_block], dtype=np.int32).tofile(f)
        np.array([ref_db], dtype=np.float32).tofile(f)
        snr.tofile(f)
        clips.tofile(f)
        irs.tofile(f)
    run = subprocess.run([EXE, str(src), str(dst)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert f"rendered {E} events x {C} capsules" in run.stdout
    out = np.fromfile(dst, dtype=np.float32)
    scale, spatial, scene = out[:E], out[E: E + E * C * La].reshape(E, C, La), out[E + E * C * La:].reshape(C, La)
    want_scene = np.zeros((C, La))
    for e in range(E):
        want = orc.render_event(clips[e], irs[:, [e], :].astype(np.float64), float(snr[e]), ref_db=ref_db, sr=sr)["spatial"]
        got = spatial[e] * scale[e]
        for c in range(C):
            assert_parity(got[c], want[c], TOL, what=(e, c))
        want_scene += want
    assert_parity(scene, want_scene, TOL)


PLANNED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c_caller", "_build", "render_planned")


@pytest.mark.parametrize("log2_block,chunk_events,lir,expect_code", [(10, 0, 2600, 612), (10, 2, 2600, 612), (13, 0, 30_001, 612),
                                                                        (13, 3, 30_001, 612), (14, 2, 60_001, 612)],
                         ids=["B1024_one_batch", "B1024_chunks_of_2", "B8192_one_batch", "B8192_chunks_of_3", "B16384_quad_tiles_chunks_of_2"])
def test_c_host_with_the_library_planner(tmp_path, log2_block, chunk_events, lir, expect_code):
    """tests/c_caller/render_planned.c: static + moving + tiled events, an ambience and a chunked batch from a C host that
    takes EVERY table from the library's planner (al_plan_create / al_plan_chunk / al_plan_emitter_parts / al_plan_mixdown);
    every row of every event and of the scene against the oracle.  The layout and accumulate flags come from the library too
    (al_plan_batch_flags: split at B = 8192, quad tiles at 16384)."""
    if not os.path.exists(PLANNED):
        import __graft_entry__

        __graft_entry__.build_c_caller()
    from tests.conftest import assert_parity

    B = 1 << log2_block
    rng = np.random.default_rng(log2_block * 10 + chunk_events)
    C, sr, ref_db, duration, amb_db = 3, 48000.0, -60.0, 9.0 * B / 48000.0 + 0.3, -55.0
    kinds = [(int(4.6 * B) + 3, 1, False), (7 * B, 12, True), (B + 5, 0, False), (3 * B - 7, 1, False), (int(5.5 * B), 9, True)]
    events, col = [], 0
    for n, ne, mv in kinds:
        events.append(dict(n=n, ne=ne, e0=col, mv=mv, snr=float(rng.uniform(5, 30)), start=float(rng.uniform(0, duration - n / sr))))
        col += ne
    N = col
    clips = [rng.standard_normal(ev["n"]).astype(np.float32) for ev in events]
    clips = [c / np.abs(c).max() for c in clips]
    irs = (rng.standard_normal((C, N, lir)) * np.exp(-np.arange(lir) / (lir / 5.0))).astype(np.float32)
    T = round(duration * sr)
    noise = rng.standard_normal((C, T)).astype(np.float32)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(src, "wb") as f:
        np.array([C, len(events), N, lir, log2_block, chunk_events, 1, 0], dtype=np.int32).tofile(f)
        np.array([ref_db, sr, duration, amb_db], dtype=np.float32).tofile(f)
        for ev in events:
            np.array([ev["n"], ev["ne"], ev["e0"], int(ev["mv"])], dtype=np.int32).tofile(f)
            np.array([ev["snr"], ev["start"]], dtype=np.float32).tofile(f)
        for c in clips:
            c.tofile(f)
        irs.tofile(f)
        noise.tofile(f)
    run = subprocess.run([PLANNED, str(src), str(dst)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr + run.stdout
    n_chunks = 1 if chunk_events == 0 else -(-len(events) // chunk_events)
    assert f"in {n_chunks} chunk(s)" in run.stdout and f"moving_code = {expect_code}" in run.stdout, run.stdout
    out = np.fromfile(dst, dtype=np.float32)
    E = len(events)
    scale, at = out[:E], E
    want_spatial = []
    for i, ev in enumerate(events):
        got = out[at: at + C * ev["n"]].reshape(C, ev["n"]).astype(np.float64) * float(scale[i])
        at += C * ev["n"]
        h = irs[:, ev["e0"]: ev["e0"] + ev["ne"], :].astype(np.float64)
        # float32(start) is what the C host was given
        want = orc.render_event(clips[i], h, float(np.float32(ev["snr"])), ref_db=ref_db, is_moving=ev["mv"], duration=ev["n"] / sr, sr=sr)["spatial"]
        for c in range(C):
            assert_parity(got[c], want[c], TOL, what=(i, c))
        want_spatial.append(want)
    scene = out[at:].reshape(C, T)
    slots = [(float(np.float32(ev["start"])), float(np.float32(ev["start"])) + ev["n"] / sr) for ev in events]
    ref = orc.mix_scene(want_spatial, slots, duration, sr, ambiences=[(orc.peak_normalise_rows(noise.astype(np.float64)), amb_db)],
                        keep_padded=False)["scene"]
    for c in range(C):
        assert_parity(scene[c], ref[c], TOL, what=("scene", c))


def test_c_host_reaches_the_benchmarked_accumulate(tmp_path):
    """The dispatch policy lives behind the C ABI (al_plan_batch_flags): on a batch in cfg2's regime (static events, 12
    partitions of 8192, clips of 13..24 blocks) the C host of render_planned.c -- which sets no flag of its own -- reaches the
    same kernels as audiblelight_amd/engine.py does from the same plan: k_spectral_mac_static<12,12,2> (code 3121202), the
    accumulate bench.py times, with the split layout; and it renders what the oracle renders."""
    if not os.path.exists(PLANNED):
        import __graft_entry__

        __graft_entry__.build_c_caller()
    import ctypes as ct

    from audiblelight_amd import _hip, engine, plan as planning

    B = 8192
    rng = np.random.default_rng(7)
    C, sr, ref_db, lir = 2, 48000.0, -65.0, 12 * B - 5
    lens = [14 * B + 3, 20 * B - 1, 23 * B + 77]
    duration = 30 * B / sr
    events = [dict(n=n, snr=float(rng.uniform(5, 30)), start=float(rng.uniform(0, duration - n / sr))) for n in lens]
    clips = [rng.standard_normal(n).astype(np.float32) for n in lens]
    clips = [c / np.abs(c).max() for c in clips]
    irs = (rng.standard_normal((C, len(lens), lir)) * np.exp(-np.arange(lir) / (lir / 5.0))).astype(np.float32)
    T = round(duration * sr)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(src, "wb") as f:
        np.array([C, len(events), len(events), lir, 0, 0, 0, 0], dtype=np.int32).tofile(f)      # log2_block 0: the library's choice
        np.array([ref_db, sr, duration, 0.0], dtype=np.float32).tofile(f)
        for i, ev in enumerate(events):
            np.array([ev["n"], 1, i, 0], dtype=np.int32).tofile(f)
            np.array([ev["snr"], ev["start"]], dtype=np.float32).tofile(f)
        for c in clips:
            c.tofile(f)
        irs.tofile(f)
    run = subprocess.run([PLANNED, str(src), str(dst)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr + run.stdout
    want_flags = _hip.FLAG_SPLIT_SPECTRA | _hip.FLAG_STATIC_MAC | _hip.FLAG_ONLY_STATIC
    assert "B = 8192, P = 12" in run.stdout and "static_code = 3121202" in run.stdout and f"flags = {want_flags}," in run.stdout, run.stdout
    # the Python host on the same plan: same flags, same instantiation
    specs = [planning.EventSpec(n_samples=ev["n"], n_emitters=1, snr=ev["snr"], emitter0=i, ref_db=ref_db) for i, ev in enumerate(events)]
    r = engine.Renderer()
    batch = r.prepare(planning.plan_batch(specs, C, lir, sr, lib=r.lib), clips, irs)
    sc, mc = ct.c_int32(), ct.c_int32()
    r.lib.call("al_spectral_mac_variant", ct.byref(batch.descs[0]), ct.byref(sc), ct.byref(mc))
    assert (sc.value, batch.descs[0].flags & ~_hip.DEBUG_FLAG_MASK) == (3121202, want_flags)
    out = np.fromfile(dst, dtype=np.float32)
    E = len(events)
    scale, at, want_spatial = out[:E], E, []
    for i, ev in enumerate(events):
        got = out[at: at + C * ev["n"]].reshape(C, ev["n"]).astype(np.float64) * float(scale[i])
        at += C * ev["n"]
        want = orc.render_event(clips[i], irs[:, [i], :].astype(np.float64), float(np.float32(ev["snr"])), ref_db=ref_db, sr=sr)["spatial"]
        assert_parity(got, want, TOL, what=i)
        want_spatial.append(want)
    slots = [(float(np.float32(ev["start"])), float(np.float32(ev["start"])) + ev["n"] / sr) for ev in events]
    ref = orc.mix_scene(want_spatial, slots, duration, sr, keep_padded=False)["scene"]
    assert_parity(out[at:].reshape(C, T), ref, TOL)
