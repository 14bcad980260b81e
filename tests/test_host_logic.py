"""Host-side logic (no GPU, no kernels): planning tables, slot rounding, chunking, C-ABI surface."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

from audiblelight_amd import _hip, plan as planning
from oracle import synth_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    """The gfx950 library loads on a machine without a GPU and exports exactly what the header declares."""
    import __graft_entry__ as entry

    entry.build()
    lib = _hip.Library(entry.LIB)
    header = open(os.path.join(ROOT, "include", "audiblelight_hip.h")).read()
    declared = set(re.findall(r"\b(al_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_hip.SYMBOLS), declared ^ set(_hip.SYMBOLS)
    assert lib.call("al_abi_version") == _hip.ABI_VERSION == 6
    assert lib.call("al_twiddle_bytes", 13) == 8 * 8192 and lib.call("al_twiddle_bytes", 9) == -1
    assert lib.call("al_row_stats_partials", 3, 40000) == 4 * 3 * 3
    assert lib.call("al_noise_workspace_floats", 2, 1000) > 0
    # argument validation happens before any launch, so it is checkable without a GPU
    with pytest.raises(_hip.HipError, match="null batch"):
        lib.call("al_render_batch", None, None)
    bad = _hip.AlBatch(log2_block=3, n_capsules=1, hop=128)
    with pytest.raises(_hip.HipError, match=r"log2_block must be in \[10, 14\]"):
        lib.call("al_ir_spectra", ct.byref(bad), None)
    # a descriptor laid out for another header version (shorter struct, older ABI) is refused, not misread
    old = _hip.AlBatch(log2_block=13, n_capsules=1, hop=128)
    old.struct_size -= 24
    with pytest.raises(_hip.HipError, match="another version of audiblelight_hip.h"):
        lib.call("al_spectral_mac", ct.byref(old), None)
    old_mix = _hip.AlMix(n_capsules=1, n_samples=8, tile=4096, n_tiles=1)
    old_mix.abi_version = 1
    with pytest.raises(_hip.HipError, match="another version of audiblelight_hip.h"):
        lib.call("al_mixdown", ct.byref(old_mix), None)


def test_struct_layouts_match_the_header():
    assert _hip.EVENT_DTYPE.itemsize == 56 and _hip.STREAM_DTYPE.itemsize == 32
    assert ct.sizeof(_hip.AlBatch) == 2 * 4 + 6 * 4 + 2 * 8 + 4 * 4 + 6 * 4 + 16 * 8 + 2 * 4 + 8   # head, ..., 16 pointers, the two zero-block indices, emitter_parts
    assert ct.sizeof(_hip.AlMix) == 2 * 4 + 6 * 4 + 13 * 8
    b, m = _hip.AlBatch(log2_block=13), _hip.AlMix()
    assert (b.struct_size, b.abi_version, m.struct_size, m.abi_version) == (232, 6, 136, 6)
    assert _hip.AlBatch.twiddle.offset % 8 == 0 and _hip.EVENT_DTYPE.fields["snr"][1] == 44


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _hip.Library(str(tmp_path / "nope.so"))


def test_event_slot_uses_bankers_rounding():
    # Python round(): 0.5 -> 0, 1.5 -> 2, 2.5 -> 2 (reference synthesize.py:361-362)
    assert planning.event_slot(0.5, 2.5, 1, 100) == (0, 2)
    assert planning.event_slot(1.5, 3.5, 1, 100) == (2, 4)
    assert planning.event_slot(-1.0, 500.0, 1, 100) == (0, 100)
    assert planning.event_slot(0.5, 2.5, 1, 100) == orc.event_slot(0.5, 2.5, 1, 100)


def test_interpolation_matrix_matches_oracle():
    for n_ir, dur, sr in ((3, 0.75, 8000), (5, 1.125, 8000), (32, 7.75, 48000), (2, 0.1, 44100)):
        w = planning.generate_interpolation_matrix(np.linspace(0, dur, n_ir), sr, 128)
        np.testing.assert_array_equal(w, orc.crossfade_weights(np.linspace(0, dur, n_ir), sr, 128))
        assert planning.stft_frame_count(int(dur * sr)) == orc.frame_count(int(dur * sr))


def test_plan_tables_static_and_moving():
    specs = [planning.EventSpec(5000, 1, 10.0, emitter0=0), planning.EventSpec(4000, 0, 5.0, emitter0=1),
             planning.EventSpec(9000, 3, 5.0, emitter0=1, is_moving=True, duration=9000 / 8000),
             planning.EventSpec(1024, 1, 5.0, emitter0=4)]
    pl = planning.plan_batch(specs, 4, 1500, 8000, log2_block=10)
    ev, st = pl.events, pl.streams
    assert list(ev["n_blocks"]) == [5, 4, 9, 1] and pl.n_partitions == 2 and pl.n_emitters == 5
    assert list(ev["n_streams"]) == [1, 0, 3, 1] and list(ev["stream0"]) == [0, 1, 2, 5]
    assert list(ev["yspec_base"]) == [0, 20, 20, 56]        # the zero-emitter event owns no spectra
    assert ev["audio_off"][1] == 5000 and ev["out_off"][1] == 4 * 5000
    assert st["w_off"][0] == -1 and st["n_j"][0] == 5 and st["gain"][2] == 512.0
    w, n_frames = orc.tv_frames(9000, 9000 / 8000, 3, 8000)
    assert ev["valid_len"][2] == min(9000, n_frames * 128 - 256) and st["w_len"][2] == n_frames
    # every non-zero sample of each envelope lies inside the planned block range of its stream
    env = orc.crossfade_envelopes(w, n_frames, 9000)
    for l in range(3):
        nz = np.flatnonzero(env[l])
        j_lo, n_j = int(st["j_lo"][2 + l]), int(st["n_j"][2 + l])
        assert (j_lo - 1) * 1024 <= nz[0] and nz[-1] < (j_lo + n_j) * 1024
    with pytest.raises(ValueError, match="Moving Event has only one emitter!"):
        planning.plan_batch([planning.EventSpec(10, 1, 1.0, is_moving=True)], 1, 10, 8000)
    with pytest.raises(ValueError, match="Expected a moving event!"):
        planning.plan_batch([planning.EventSpec(10, 2, 1.0)], 1, 10, 8000)
    with pytest.raises(ValueError, match="win_size == 2"):
        planning.plan_batch(specs, 4, 1500, 8000, hop=100, win=256)


def test_chunks_partition_the_tables():
    specs = [planning.EventSpec(3000 + 100 * i, 1, 10.0, emitter0=i) for i in range(7)]
    pl = planning.plan_batch(specs, 3, 2000, 8000, log2_block=10)
    chunks = pl.chunks(3)
    assert [c["n_events"] for c in chunks] == [3, 3, 1]
    assert sum(c["xspec_blocks"] for c in chunks) == pl.xspec_blocks
    assert sum(c["yspec_blocks"] for c in chunks) == pl.yspec_blocks
    assert [c["emitter0"] for c in chunks] == [0, 3, 6] and all(c["n_emitters"] == c["n_events"] for c in chunks)
    assert pl.chunks(None)[0]["n_events"] == 7


def test_mix_plan_tiles_and_clipping():
    # event 2 runs past the scene end, event 1 is empty after rounding, event 3 is longer than its slot
    mix = planning.plan_mixdown(starts=[0.1, 5.0, 1.9, 0.0], ends=[0.6, 5.5, 2.6, 0.25], lens=[4000, 100, 5600, 9000],
                                rows=[4, 4, 4, 2], src_offsets=[0, 16000, 16400, 38800], event_index=[0, 1, 2, 3],
                                duration=2.0, sample_rate=8000, n_capsules=4, tile=4096)
    assert mix.skipped == [1] and mix.n_samples == 16000 and mix.n_tiles == 4
    assert list(mix.slot_start) == [800, 15200, 0] and list(mix.slot_count) == [4000, 800, 2000]
    assert list(mix.slot_event) == [0, 2, 3] and list(mix.slot_rows) == [4, 4, 2]
    lists = [list(mix.tile_events[mix.tile_ptr[t]: mix.tile_ptr[t + 1]]) for t in range(4)]
    assert lists == [[0, 2], [0], [], [1]]


def test_block_size_choice():
    assert planning.choose_log2_block(96000, 192000) == 13
    assert planning.choose_log2_block(1000, 6000) == 10
    assert planning.choose_log2_block(5000, 3000) == 12


def test_bench_cpu_baseline_leg_runs_on_a_small_sample():
    """bench.py's cpu_baseline (the oracle timed on host cores) on a shrunken cfg2 and cfg3 scene: keys and sanity."""
    import bench
    from audiblelight_amd import synthetic

    for name, kw in (("cfg2", dict(scale=0.02)), ("cfg3", dict(scale=0.02, E=2))):
        sc = synthetic.make_scene(name, **kw)
        out = bench.cpu_baseline(sc, 2)
        assert out["kind"] == "port" and out["cores"] == 1 and out["unit"] == "scene-seconds/s"
        assert out["value"] > 0 and "events" in out["sample"] and out["cpu_model"]


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` starts two rank processes itself (gloo + host-emulated kernels here; RCCL + gfx950 on
    the GPU box) and rank 0 prints ONE JSON line with n_gpus == 2, a SELF-VALIDATED gather and the per-step roofline
    fields; the scene-batch mode (cfg4's contract) and the capsule-sharded mode (cfg5's secondary contract) likewise."""
    import json
    import subprocess
    import sys

    from tests import hostemu

    hostemu.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(AL_BENCH_EMULATE="1", AL_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "cfg1",
           "--scale", "0.05", "--cpu-events", "0"]
    def run(extra):
        res = subprocess.run(cmd + extra, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    out = run([])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert 0 < out["roofline"]["path_frac"] <= out["roofline"]["frac"]
    assert "HOST EMULATION" in out["data"]
    assert out["timing"]["repeats"] == 3 and len(out["timing"]["ms_per_step_each_repeat"]) == 3
    assert len(out["timing"]["ms_per_step_by_rank_last_repeat"]) == 2
    # the first multi-rank run validates itself: rank 0 renders rank 1's scene again and compares the gathered buffer bit for bit
    g = out["gather"]
    assert g["bytes_total"] > 0 and g["ranks_seen"] == [0, 1] and g["validated_against_local_rerender"] == {"1": True} and g["bit_exact"]
    # the collection held against the rendering and against one xGMI link: the line says whether the links or the kernels set the pace
    assert g["xgmi_link_peak_GBps"] == 153.0 and g["link_frac"] > 0 and g["render_ms_over_gather_ms"] > 0
    assert g["link_GBps_needed_to_keep_up"] > 0 and g["render_ms"] == out["ms_per_step"]
    # BASELINE configs[3] as a mode: a batch of scenes split over the ranks, ALL of them gathered and two of them re-rendered
    out = run(["--total-scenes", "4", "--repeats", "1"])
    assert out["scaling"] == "strong" and out["config"]["total_scenes"] == 4 and out["config"]["scenes_this_rank"] == 2
    assert out["gather"]["validated_against_local_rerender"] == {"1": True, "3": True} and out["gather"]["bit_exact"]
    assert out["gather"]["bytes_total"] == 4 * (4 * 12000 * 4)      # ALL four (4 capsules x 12 000 samples) scenes arrived
    ov = out["gather"]["overlapped"]                                # the same collection overlapped with one more step of rendering
    assert ov["bit_exact"] and ov["step_with_gather_ms"] > 0 and "gather_overlap_ms" in ov
    # SURVEY 8e row 2 as a mode: ONE scene, capsule rows split over the ranks, two all-reduces inside the step
    out = run(["--shard", "capsules", "--repeats", "1"])
    assert out["config"]["shard"] == "capsules" and out["config"]["capsules_this_rank"] == 2 and out["scaling"] == "strong"
    assert set(out["collectives_ms"]) >= {"allreduce_ir_norms", "allreduce_event_levels"}
    assert out["gather"]["rows_total"] == 4 and out["gather"]["within_tolerance"]
    # a launcher that sets WORLD_SIZE differently from --gpus is an error, not a silent 1-GPU run
    bad = subprocess.run(cmd, env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in bad.stderr


def test_bench_line_carries_parity_of_what_it_timed():
    """The single-rank bench line compares the scene buffer its timed steps wrote with the oracle's mix of ALL events (the
    scene the cpu_baseline leg renders anyway): `parity` = both halves of the 1e-4 contract over every capsule row and sample,
    exit code 1 above tolerance; a bounded `--parity-events` goes through a GPU mixdown of those events alone.  Host-emulated
    kernels here (the arithmetic is the same C++ code compiled for the CPU), the gfx950 build on the GPU box.  A non-default
    A/B switch shows up in config.switches."""
    import json
    import subprocess
    import sys

    from tests import hostemu

    hostemu.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK") and not k.startswith("AL_")}
    env.update(AL_BENCH_EMULATE="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--repeats", "1", "--config", "cfg2",
           "--scale", "0.005", "--cpu-workers", "0", "--end-to-end", "0", "--dropin", "0"]

    def run(extra, **more_env):
        res = subprocess.run(cmd + extra, env=dict(env, **more_env), capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])

    out = run(["--cpu-events", "64"])
    par = out["parity"]
    assert par["ok"] and par["events"] == 64 and par["rows"] == 32 and par["samples"] == round(60 * 0.005 * 48000)
    assert 0 < par["rel_rms"] < 1e-5 and 0 < par["max_abs_over_peak"] < 1e-5 and par["tol"] == 1e-4
    assert "timed steps wrote" in par["note"] and out["config"]["switches"] == {}
    assert out["cpu_baseline"]["extrapolated"] is False
    out = run(["--cpu-events", "2", "--parity-events", "5"], AL_STATIC_MAC="0")
    # five events in full (all rows x all samples), and one pseudo-random capsule row of EACH of the other 59: every event meets the oracle
    par = out["parity"]
    assert par["ok"] and par["events_in_full"] == 5 and par["events"] == 64 and "first 5 of 64" in par["note"]
    assert par["rows_sampled"]["events"] == 59 and par["rows_sampled"]["ok"] and 0 < par["rows_sampled"]["rel_rms_worst_row"] < 1e-5
    assert out["config"]["switches"] == {"static_mac": False}


def test_bench_eight_ranks_scene_batch_index_arithmetic():
    """BASELINE configs[3]'s launch shape -- eight ranks, a batch of scenes split round-robin, rank 0 receiving from SEVEN peers --
    on gloo with the host-emulated kernels at a reduced scene size: 19 scenes (uneven shares: three ranks own 3, five own 2),
    every one gathered at the end AND once more overlapped with the rendering (receives posted per round up front), the
    gathered buffers of scenes 1 and 18 compared bit for bit with rank 0's own re-render; and the N = 1 run of the same command
    makes the same C-ABI calls per step as every rank of the N = 8 run."""
    import json
    import subprocess
    import sys

    from tests import hostemu

    hostemu.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(AL_BENCH_EMULATE="1", AL_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--repeats", "1", "--config", "cfg1",
            "--scale", "0.05", "--cpu-events", "0", "--cpu-workers", "0", "--end-to-end", "0", "--dropin", "0"]

    def run(extra):
        res = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    out = run(["--gpus", "8", "--total-scenes", "19"])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["total_scenes"] == 19 and out["config"]["scenes_this_rank"] == 3
    g = out["gather"]
    assert g["ranks_seen"] == list(range(8)) and g["bytes_total"] == 19 * (4 * 12000 * 4)
    assert g["validated_against_local_rerender"] == {"1": True, "18": True} and g["bit_exact"] and g["overlapped"]["bit_exact"]
    assert len(out["timing"]["ms_per_step_by_rank_last_repeat"]) == 8
    # every rank took its share of the host: helper pools capped at usable CPUs / 8, the process pinned to that many CPUs
    cpus = len(os.sched_getaffinity(0))
    hs = out["host_share"]
    assert hs["threads_by_rank"] == [max(1, cpus // 8)] * 8 and hs["pinned"] and hs["cpus"] == max(1, cpus // 8)
    assert g["bytes_over_links"] == (19 - 3) * (4 * 12000 * 4) and g["GBps_into_root"] > 0 and g["GBps_per_peer_link"] > 0
    weak8, one = run(["--gpus", "8"]), run(["--gpus", "1"])
    assert weak8["n_gpus"] == 8 and weak8["gather"]["ranks_seen"] == list(range(8)) and weak8["gather"]["bit_exact"]
    assert weak8["gather"]["validated_against_local_rerender"] == {"1": True, "7": True}
    assert weak8["config"]["step_calls"] == one["config"]["step_calls"] and one["n_gpus"] == 1
    assert weak8["config"]["workload"] == one["config"]["workload"]
    # SURVEY 8e row 2 at eight ranks: one cfg2-shaped scene (32 capsules, shrunk lengths), four capsule rows per rank, the two
    # all-reduces inside the step, rows gathered and compared with rank 0's render of the whole scene
    caps = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--repeats", "1",
                           "--config", "cfg2", "--scale", "0.02", "--shard", "capsules", "--cpu-events", "0", "--cpu-workers", "0"],
                          env=env, capture_output=True, text=True, timeout=900)
    assert caps.returncode == 0, caps.stderr[-3000:]
    out = json.loads([ln for ln in caps.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["config"]["capsules_this_rank"] == 4 and out["gather"]["rows_total"] == 32
    assert out["gather"]["within_tolerance"] and out["gather"]["ranks_seen"] == list(range(8))


def test_header_is_plain_c_and_the_c_host_links():
    """include/audiblelight_hip.h must be usable from C (the boundary is a C ABI, not a C++ one): the plain-C host of
    tests/c_caller compiles with gcc -std=c11 against it and links every entry point it calls (no GPU needed to link);
    the struct sizes gcc sees are the ones the ctypes mirror uses."""
    import ctypes as ct
    import subprocess
    import tempfile

    import __graft_entry__
    from audiblelight_amd import _hip

    exe = __graft_entry__.build_c_caller()
    assert os.path.exists(exe)
    probe = r'''
#include <stdio.h>
#include "audiblelight_hip.h"
int main(void) { printf("%zu %zu %zu %zu\n", sizeof(al_event), sizeof(al_stream), sizeof(al_batch), sizeof(al_mix)); return 0; }
'''
    with tempfile.TemporaryDirectory() as tmp:
        src, out = os.path.join(tmp, "sizes.c"), os.path.join(tmp, "sizes")
        with open(src, "w") as f:
            f.write(probe)
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", out])
        sizes = [int(x) for x in subprocess.check_output([out], text=True).split()]
    assert sizes == [_hip.EVENT_DTYPE.itemsize, _hip.STREAM_DTYPE.itemsize, ct.sizeof(_hip.AlBatch), ct.sizeof(_hip.AlMix)]


def test_shard_stream_is_lazy_round_robin():
    """distributed.shard_stream: rank r of w gets items r, r + w, ... of a generator; the ranks cover it disjointly."""
    from audiblelight_amd import distributed

    seen = [list(distributed.shard_stream(iter(range(11)), r, 3)) for r in range(3)]
    assert seen == [[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8]]
    assert sorted(sum(seen, [])) == list(range(11))
    import pytest as _pytest

    with _pytest.raises(ValueError):
        list(distributed.shard_stream(range(3), 2, 2))


def test_planner_invariants_on_random_scenes():
    """Property test of the host planner over random event lists: workspace regions are disjoint and 16-byte aligned, every
    tile of the mixdown plan lists exactly the events that overlap it, in insertion order, and the per-tile work adds up to
    what a plain loop over the events would add (reference loop: synthesize.py:358-383)."""
    from hypothesis import given, settings, strategies as st

    from audiblelight_amd import plan as planning

    event = st.tuples(st.integers(1, 40_000), st.sampled_from([0, 1, 1, 1, 3, 6]), st.floats(0.0, 9.0), st.floats(-2.0, 30.0))

    @settings(max_examples=250, deadline=None)
    @given(st.lists(event, min_size=1, max_size=7), st.integers(1, 5), st.integers(1, 9000), st.sampled_from([10, 11, 12, 13]))
    def check(events, n_capsules, ir_len, log2_block):
        sr, duration = 8000, 10.0
        specs, col = [], 0
        for n, n_emit, _, snr in events:
            specs.append(planning.EventSpec(n_samples=n, n_emitters=n_emit, snr=snr, emitter0=col, is_moving=n_emit > 1,
                                            duration=n / sr if n_emit > 1 else None))
            col += n_emit
        pl = planning.plan_batch(specs, n_capsules, ir_len, sr, log2_block=log2_block)
        ev = pl.events
        B = 1 << log2_block
        assert pl.n_emitters == col and pl.n_partitions in (-(-ir_len // B), 0 if col == 0 else -1)
        # clips and event outputs: 16-byte aligned, back to back, no overlap
        assert (ev["audio_off"] % 4 == 0).all() and (ev["out_off"] % 4 == 0).all()
        for a, b in zip(range(len(ev) - 1), range(1, len(ev))):
            assert ev["audio_off"][a] + ev["len"][a] <= ev["audio_off"][b]
            assert ev["out_off"][a] + n_capsules * ev["len"][a] <= ev["out_off"][b]
        assert pl.audio_floats >= ev["audio_off"][-1] + ev["len"][-1] and pl.spatial_floats >= ev["out_off"][-1] + n_capsules * ev["len"][-1]
        assert (ev["n_blocks"] == -(-ev["len"] // B)).all() and (ev["valid_len"] <= ev["len"]).all()
        # output-spectra and statistics regions of consecutive events do not overlap
        conv = ev["n_streams"] > 0
        y_end = ev["yspec_base"] + n_capsules * ev["n_blocks"] * conv
        assert (ev["yspec_base"][1:] >= y_end[:-1]).all() and pl.yspec_blocks >= int(y_end.max())
        assert (ev["part_base"][1:] == ev["part_base"][:-1] + n_capsules * ev["n_blocks"][:-1]).all()
        # signal-spectra runs of the streams are disjoint and inside the clip's blocks
        runs = sorted((int(s["xspec_base"]), int(s["n_j"])) for s in pl.streams if s["n_j"] > 0)
        for (a0, an), (b0, _) in zip(runs, runs[1:]):
            assert a0 + an <= b0
        for s in pl.streams:
            assert 0 <= s["j_lo"] and s["j_lo"] + s["n_j"] <= max(int(ev["n_blocks"][s["event"]]), 1)
        # mixdown plan against a plain loop
        starts = [e[2] for e in events]
        ends = [s0 + n / sr for s0, (n, *_r) in zip(starts, events)]
        lens = [e[0] for e in events]
        mp = planning.plan_mixdown(starts, ends, lens, [n_capsules] * len(events), ev["out_off"], list(range(len(events))),
                                   duration, sr, n_capsules)
        n_scene = round(duration * sr)
        want_cover = np.zeros(n_scene, dtype=np.int64)
        kept = []
        for i, (t0, t1, la) in enumerate(zip(starts, ends, lens)):
            a, b = planning.event_slot(t0, t1, sr, n_scene)
            if b > a:
                kept.append(i)
                want_cover[a: a + min(b - a, la)] += 1
        assert sorted(kept + list(mp.skipped)) == list(range(len(events))) and list(mp.slot_event[: len(kept)]) == kept
        got_cover = np.zeros(n_scene, dtype=np.int64)
        for t in range(mp.n_tiles):
            slots = mp.tile_events[mp.tile_ptr[t]: mp.tile_ptr[t + 1]]
            assert list(slots) == sorted(slots)                                   # insertion order inside a tile
            lo, hi = t * mp.tile, min((t + 1) * mp.tile, n_scene)
            for sl in slots:
                a, cnt = int(mp.slot_start[sl]), int(mp.slot_count[sl])
                assert a < hi and a + cnt > lo                                    # the slot really overlaps the tile
                got_cover[max(a, lo): min(a + cnt, hi)] += 1
        np.testing.assert_array_equal(got_cover, want_cover)

    check()


def test_planner_block_size_rule_for_long_irs():
    """al_plan_create picks B = 16384 (csrc/al_quad16.h) only where it was measured to pay (profiles/r04u_quad16_ir_sweep_*.txt): all
    events static, 17..24 partitions of 8192, at least 100 000 (capsule, block) rows; the numpy planner of rounds 1-3 with the same
    rule agrees table for table."""
    from tests import plan_reference as ref

    def both(n_events, C, ir_len, n_samples=192000, moving_at=None):
        kw = [dict(n_samples=n_samples, n_emitters=1, snr=5.0, emitter0=e) for e in range(n_events)]
        if moving_at is not None:
            kw[moving_at] = dict(n_samples=n_samples, n_emitters=4, snr=5.0, emitter0=n_events, is_moving=True, duration=n_samples / 48000.0)
        a = planning.plan_batch([planning.EventSpec(**k) for k in kw], C, ir_len, 48000.0)
        b = ref.plan_batch([ref.EventSpec(**k) for k in kw], C, ir_len, 48000.0)
        assert a.log2_block == b.log2_block and a.events.tobytes() == b.events.tobytes() and a.streams.tobytes() == b.streams.tobytes()
        return a

    assert (both(128, 64, 192000).log2_block, both(128, 64, 192000).n_partitions) == (14, 12)       # BASELINE configs[4]
    assert both(70, 64, 8192 * 16 + 1).log2_block == 14                                             # 17 partitions, 107 520 rows
    assert both(65, 64, 8192 * 16 + 1).log2_block == 13                                             # 99 840 rows
    assert both(128, 64, 8192 * 16).log2_block == 13                                                # 16 partitions
    assert both(128, 64, 8192 * 24 + 1).log2_block == 13                                            # 25 partitions
    assert both(32, 32, 192000).log2_block == 13                                                    # small batch
    assert both(128, 64, 192000, moving_at=5).log2_block == 13                                      # a moving event on the sliding window at 8192: rule (2) keeps it
    assert planning.plan_batch([planning.EventSpec(192000, 1, 5.0)] * 128, 64, 192000, 48000.0, log2_block=13).log2_block == 13   # the caller's choice stands


def test_planner_block_size_rule_for_moving_events():
    """al_plan_create, rule (2): a batch with moving events gets B = 16384 exactly when some moving event is off the sliding-window
    accumulate at B = 8192 (a cross-fade window of more than AL_SPARSE_MAX_NJ blocks, or more than AL_SPARSE_MAX_PARTITIONS
    partitions) and all of them are on it at 16384 (profiles/r04z_quad16_moving_sweep*.txt); the numpy planner agrees."""
    from tests import plan_reference as ref

    def both(n_irs, ir_len, n_samples=372000, with_static=False):
        kw = [dict(n_samples=n_samples, n_emitters=n_irs, snr=5.0, emitter0=0, is_moving=True, duration=n_samples / 48000.0)]
        if with_static:
            kw.append(dict(n_samples=100000, n_emitters=1, snr=5.0, emitter0=n_irs))
        a = planning.plan_batch([planning.EventSpec(**k) for k in kw], 4, ir_len, 48000.0)
        b = ref.plan_batch([ref.EventSpec(**k) for k in kw], 4, ir_len, 48000.0)
        assert a.log2_block == b.log2_block and a.events.tobytes() == b.events.tobytes() and a.streams.tobytes() == b.streams.tobytes()
        return a

    pl = both(32, 96000)                       # cfg3: windows of 2.84 blocks, 12 partitions: on the sliding window at 8192
    assert pl.log2_block == 13 and int(pl.events["reserved"][0]) == 1
    pl = both(32, 192000)                      # 24 partitions: still eligible
    assert pl.log2_block == 13
    pl = both(32, 200000)                      # 25 partitions of 8192: only 16384 keeps the event on the sliding window
    assert pl.log2_block == 14 and pl.n_partitions == 13 and int(pl.events["reserved"][0]) == 1
    pl = both(16, 96000)                       # 16 waypoints: windows of 5.7 blocks of 8192 reach 7-8 signal blocks, 4 at 16384
    assert pl.log2_block == 14 and int(pl.events["reserved"][0]) == 1
    assert both(16, 96000, with_static=True).log2_block == 14
    pl = both(4, 96000)                        # windows of 23 blocks: off it at either size -> the default stays
    assert pl.log2_block == 13 and int(pl.events["reserved"][0]) == 0
    assert both(32, 8192 * 48 + 1).log2_block == 13        # 25 partitions of 16384: neither


def test_every_accumulate_instantiation_is_named_by_a_gpu_test():
    """Every k_spectral_mac* kernel the library holds (nm -C) must be the expected instantiation of some -m gpu parity test
    (tests/mac_regimes.py: STATIC_CASES, STATIC_LOOP_CASES, MOVING_CODES, EXTRA_STATIC_CODES, each asserted through
    al_spectral_mac_variant, which reads the launcher's own descriptor): no accumulate kernel ships unpinned."""
    import subprocess

    from tests import mac_regimes as mr

    out = subprocess.check_output(["nm", "-C", _hip.DEFAULT_LIB]).decode()
    syms = {line.split(" ", 2)[2] for line in out.splitlines()
            if "k_spectral_mac" in line and "__device_stub__" not in line}
    assert len(syms) >= 40                                  # 12 x {one k-tile, pair, LDS ring} + 2 two-unit + tile + moving kernels
    asserted, unpinned = mr.asserted_codes(), []
    for sym in sorted(syms):
        codes = mr.codes_of_kernel_symbol(sym)
        assert codes, f"cannot map {sym} to a variant code"
        unpinned += [(sym, c) for c in codes if c not in asserted]
    assert not unpinned, unpinned
    # and the other way round: every asserted code has a kernel behind it
    have = {c for sym in syms for c in mr.codes_of_kernel_symbol(sym)}
    assert asserted <= have, sorted(asserted - have)


# ----------------------------------------------------------------------------- the planner behind the C ABI (csrc/al_plan.cpp)
def _random_specs(rng, sr):
    specs, col = [], 0
    for _ in range(int(rng.integers(1, 6))):
        kind = rng.choice(["static", "static", "moving", "dry"])
        n = int(rng.integers(600, 200000)) if kind == "moving" else int(rng.integers(1, 200000))
        ne = {"static": 1, "dry": 0, "moving": int(rng.integers(2, 40))}[kind]
        specs.append(dict(n_samples=n, n_emitters=ne, snr=float(rng.uniform(5, 30)), emitter0=col, is_moving=ne > 1, duration=n / sr,
                          gain=float(rng.uniform(0.1, 2)), ref_db=float(rng.uniform(-70, -40))))
        col += ne
    return specs


def _same_optional(a, b):
    return (a is None and b is None) or (a is not None and b is not None and np.array_equal(a, b))


@pytest.mark.parametrize("seed", range(6))
def test_c_planner_tables_equal_the_numpy_planner(seed):
    """al_plan_create / al_plan_emitter_parts / al_workspace_bytes (what audiblelight_amd/plan.py calls) against the numpy
    planner of rounds 1-3 (tests/plan_reference.py): every table bit for bit on random batches of static, moving and tiled
    events, every block size."""
    from tests import plan_reference as ref

    rng = np.random.default_rng(900 + seed)
    for _ in range(40):
        sr = float(rng.choice([8000, 16000, 44100, 48000]))
        C, L = int(rng.integers(1, 9)), int(rng.integers(1, 40000))
        lb = rng.choice([0, 10, 11, 12, 13, 14])
        lb = None if lb == 0 else int(lb)
        specs = _random_specs(rng, sr)
        a = planning.plan_batch([planning.EventSpec(**k) for k in specs], C, L, sr, log2_block=lb)
        b = ref.plan_batch([ref.EventSpec(**k) for k in specs], C, L, sr, log2_block=lb)
        assert a.log2_block == b.log2_block and a.events.tobytes() == b.events.tobytes() and a.streams.tobytes() == b.streams.tobytes()
        assert np.array_equal(a.wtab, b.wtab) and np.array_equal(a.audio_offsets, b.audio_offsets)
        assert (a.audio_floats, a.spatial_floats, a.xspec_blocks, a.yspec_blocks, a.n_partials, a.n_emitters) == \
               (b.audio_floats, b.spatial_floats, b.xspec_blocks, b.yspec_blocks, b.n_partials, b.n_emitters)
        pa, pb = a.emitter_parts(), b.emitter_parts()
        assert _same_optional(pa, pb)
        assert a.workspace_bytes() == b.workspace_bytes() and a.max_nj_sliding() == b.max_nj_sliding()
        assert _hip.get_library().call("al_workspace_bytes", a._c_plan()) == b.workspace_bytes()   # the plan owns its handle


def test_c_planner_mixdown_and_weights_equal_the_numpy_planner():
    from tests import plan_reference as ref

    rng = np.random.default_rng(77)
    for _ in range(200):
        sr = float(rng.choice([8000, 22050, 48000]))
        dur = float(rng.uniform(0.2, 20.0))
        n = int(rng.integers(0, 9))
        starts = rng.uniform(-1.0, dur, n)
        lens = rng.integers(1, 200000, n).astype(np.int32)
        ends = starts + lens / sr + rng.uniform(-0.5, 0.5, n)
        # half-sample starts exercise Python's round-half-even (synthesize.py:361-362)
        if n:
            starts[0] = (int(starts[0] * sr) + 0.5) / sr
        rows = rng.integers(1, 5, n).astype(np.int32)
        src = np.cumsum(rng.integers(4, 1000, n)).astype(np.int64)
        a = planning.plan_mixdown(starts, ends, lens, rows, src, list(range(n)), dur, sr, 4)
        b = ref.plan_mixdown(starts, ends, lens, rows, src, list(range(n)), dur, sr, 4)
        for f in ("tile_ptr", "tile_events", "slot_src", "slot_len", "slot_start", "slot_count", "slot_rows", "slot_event"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert (a.n_samples, a.n_tiles, a.skipped) == (b.n_samples, b.n_tiles, b.skipped)
    for n_ir, dur, sr in ((3, 0.75, 8000), (5, 1.125, 8000), (32, 7.75, 48000), (2, 0.1, 44100), (17, 3.3333, 22050), (1, 1.0, 8000)):
        t = np.linspace(0, dur, n_ir)
        np.testing.assert_array_equal(planning.generate_interpolation_matrix(t, sr, 128), ref.generate_interpolation_matrix(t, sr, 128))
        np.testing.assert_array_equal(planning.generate_interpolation_matrix(t, sr, 128), orc.crossfade_weights(t, sr, 128))


def test_c_planner_refuses_what_the_reference_refuses():
    with pytest.raises(ValueError, match="Moving Event has only one emitter!"):
        planning.plan_batch([planning.EventSpec(10, 1, 1.0, is_moving=True)], 1, 10, 8000)
    with pytest.raises(ValueError, match="Expected a moving event!"):
        planning.plan_batch([planning.EventSpec(10, 2, 1.0)], 1, 10, 8000)
    with pytest.raises(ValueError, match="at least one sample"):
        planning.plan_batch([planning.EventSpec(0, 1, 1.0)], 1, 10, 8000)
    with pytest.raises(ValueError, match="moving events need Event.duration"):
        planning.plan_batch([planning.EventSpec(9000, 3, 1.0, is_moving=True)], 1, 10, 8000)


def test_c_planner_edge_cases():
    """Empty batches, a scene without events, chunk ranges outside the plan, an event whose slot is empty after rounding."""
    lib = _hip.get_library()
    empty = planning.plan_batch([], 3, 1000, 8000)
    assert len(empty.events) == 0 and empty.n_emitters == 0 and empty.n_partitions == 0 and empty.workspace_bytes() > 0
    assert empty.emitter_parts() is None and empty.chunks(2)[0]["n_events"] == 0
    pl = planning.plan_batch([planning.EventSpec(3000, 1, 5.0), planning.EventSpec(2000, 0, 5.0, emitter0=1)], 2, 500, 8000, log2_block=10)
    ch = _hip.AlChunk()
    with pytest.raises(_hip.HipError, match="bad chunk range"):
        lib.call("al_plan_chunk", pl._c_plan(), 1, 5, ct.byref(ch))
    flags = ct.c_int32(0)
    for e0, n in ((1, 2**31 - 1), (2**31 - 1, 1), (3, 0), (-1, 1)):      # sums past INT32_MAX must not wrap into range
        with pytest.raises(_hip.HipError, match="bad chunk range"):
            lib.call("al_plan_chunk", pl._c_plan(), e0, n, ct.byref(ch))
        with pytest.raises(_hip.HipError, match="bad chunk range"):
            lib.call("al_plan_batch_flags", pl._c_plan(), ct.byref(_hip.AlChunk(event0=e0, n_events=n)), ct.byref(flags))
    lib.call("al_plan_chunk", pl._c_plan(), 1, 1, ct.byref(ch))          # a chunk of one tiled event: no spectra of its own
    assert (ch.n_emitters, ch.xspec_blocks, ch.yspec_blocks, ch.n_streams, ch.max_blocks) == (0, 0, 0, 1, 2)
    mix = planning.plan_mixdown([], [], [], [], [], [], 1.0, 8000, 2)
    assert mix.n_samples == 8000 and mix.n_tiles == 2 and list(mix.tile_ptr) == [0, 0, 0] and mix.skipped == []
    mix = planning.plan_mixdown([0.99999, 0.5], [1.5, 0.50001], [4000, 10], [2, 2], [0, 8000], [0, 1], 1.0, 8000, 2)
    assert mix.skipped == [0, 1] and list(mix.tile_ptr) == [0, 0, 0]   # both slots are empty after rounding (synthesize.py:364-370)
    assert lib.call("al_choose_log2_block", 96000, 192000) == 13 and lib.call("al_stft_frame_count", 9000, 128) == 73


def test_dispatch_policy_lives_behind_the_c_abi():
    """al_plan_batch_flags (csrc/al_plan.cpp) against an independent statement of the rule -- split layout at B = 8192, + quad
    tiles at 16384, capsule-loop accumulate for one-emitter events up to 21 partitions, ONLY_STATIC for chunks without a
    multi-emitter event -- on random plans and chunkings; and the descriptors engine.Renderer.prepare builds carry exactly those
    flags (no policy left in Python: switches.current() is all-default here)."""
    from audiblelight_amd import engine, plan as planning, switches
    from tests import hostemu

    assert switches.current().non_default() == {} and not switches.current().forces_dispatch
    lib = _hip.Library(hostemu.build())
    r = engine.Renderer(lib=lib, memory=hostemu.NumpyMemory())
    rng = np.random.default_rng(11)
    seen = set()
    for trial in range(60):
        lb = int(rng.choice([10, 11, 12, 13, 14]))
        B = 1 << lb
        n_ev = int(rng.integers(1, 6))
        P = int(rng.choice([1, 3, 12, 13, 21, 22, 25])) if lb <= 11 else int(rng.integers(1, 4))
        ir_len = P * B - int(rng.integers(0, B // 2))
        specs, col = [], 0
        for _ in range(n_ev):
            ne = int(rng.choice([0, 1, 1, 1, 3]))
            n = int(rng.integers(B // 2, 3 * B))
            specs.append(planning.EventSpec(n_samples=n, n_emitters=ne, snr=10.0, emitter0=col, is_moving=ne > 1, duration=n / 48000.0 if ne > 1 else None))
            col += ne
        if col == 0:
            continue
        pl = planning.plan_batch(specs, 2, ir_len, 48000.0, log2_block=lb, lib=lib)
        step = int(rng.integers(1, n_ev + 1))

        def want(chunk):
            ev = pl.events[chunk["event0"]: chunk["event0"] + chunk["n_events"]]
            f = {13: _hip.FLAG_SPLIT_SPECTRA, 14: _hip.FLAG_SPLIT_SPECTRA | _hip.FLAG_QUAD_SPECTRA}.get(lb, 0)
            if pl.n_partitions <= 21 and (ev["n_streams"] == 1).any():
                f |= _hip.FLAG_STATIC_MAC | (0 if (ev["n_streams"] > 1).any() else _hip.FLAG_ONLY_STATIC)
            return f

        chunks = pl.chunks(step)
        for ch in chunks:
            assert pl.batch_flags(ch) == want(ch), (trial, ch)
            seen.add(want(ch))
        whole = dict(event0=0, n_events=n_ev)
        assert pl.batch_flags(None) == want(whole)
        if lb <= 11 and trial % 4 == 0:      # the descriptors of a prepared batch (host emulation: small blocks only)
            clips = [rng.standard_normal(sp.n_samples).astype(np.float32) for sp in specs]
            irs = rng.standard_normal((2, col, ir_len)).astype(np.float32)
            batch = r.prepare(pl, clips, irs, chunk_events=step)
            for desc, ch in zip(batch.descs, chunks):
                assert desc.flags == want(ch)
    assert len(seen) >= 5, seen
    with pytest.raises(_hip.HipError):
        ch = _hip.AlChunk(event0=3, n_events=99)
        import ctypes as ct

        lib.call("al_plan_batch_flags", pl._c_plan(), ct.byref(ch), ct.byref(ct.c_int32()))


def test_planning_needs_neither_rocm_nor_torch():
    """csrc/al_plan.cpp is linked into a library of its own (libaudiblelight_plan.so: plain C++, no HIP): planning -- tables, workspace
    sizes, block size, dispatch flags, mixdown slots -- runs in a process that never loads the HIP library, never imports torch and
    could not find the GPU library if it tried; and it gives the tables the full library gives."""
    import subprocess
    import sys

    code = r'''
import os, sys
os.environ["AUDIBLELIGHT_HIP_LIB"] = "/nonexistent/libaudiblelight_hip.so"
from audiblelight_amd import _hip, plan as planning
specs = [planning.EventSpec(n_samples=50_000, n_emitters=1, snr=10.0, emitter0=0),
         planning.EventSpec(n_samples=70_000, n_emitters=4, snr=12.0, emitter0=1, is_moving=True, duration=70_000 / 48000.0)]
pl = planning.plan_batch(specs, 3, 30_001, 48000.0)
mp = planning.plan_mixdown([0.1, 0.5], [1.2, 2.0], [50_000, 70_000], [3, 3], pl.events["out_off"], [0, 1], 2.5, 48000.0, 3)
assert len(pl.chunks(1)) == 2
assert "torch" not in sys.modules and _hip._default is None and _hip.get_planner().path.endswith("libaudiblelight_plan.so")
print(int(pl.events["yspec_base"][1]), pl.xspec_blocks, int(mp.tile_ptr[-1]), planning.stft_frame_count(1000), pl.log2_block, pl.batch_flags())
'''
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert res.returncode == 0, res.stderr[-2000:]
    from audiblelight_amd import plan as planning

    specs = [planning.EventSpec(n_samples=50_000, n_emitters=1, snr=10.0, emitter0=0),
             planning.EventSpec(n_samples=70_000, n_emitters=4, snr=12.0, emitter0=1, is_moving=True, duration=70_000 / 48000.0)]
    full = planning.plan_batch(specs, 3, 30_001, 48000.0, lib=_hip.get_library())
    mp = planning.plan_mixdown([0.1, 0.5], [1.2, 2.0], [50_000, 70_000], [3, 3], full.events["out_off"], [0, 1], 2.5, 48000.0, 3,
                               lib=_hip.get_library())
    assert res.stdout.split() == [str(int(full.events["yspec_base"][1])), str(full.xspec_blocks), str(int(mp.tile_ptr[-1])), "9",
                                  str(full.log2_block), str(full.batch_flags())]
    assert full.batch_flags() & _hip.FLAG_STATIC_MAC and not full.batch_flags() & _hip.FLAG_ONLY_STATIC


def test_end_to_end_record_aggregates_over_concurrent_ranks():
    """The PCIe-inclusive leg under N > 1 (every rank runs it at the same time): `value` is ALL ranks' scene-seconds over the SLOWEST
    rank's wall time of the best pass, per-rank rates and the aggregate host traffic sit beside it; with one rank the record is the
    single-GPU one."""
    import types

    import bench
    from audiblelight_amd.batch import BatchReport

    scene = types.SimpleNamespace(irs=np.zeros((2, 3, 1000), np.float32))
    reps = [BatchReport(n_scenes=4, scene_seconds=240.0, wall_s=w, h2d_bytes=4 * 800, d2h_bytes=4 * 300) for w in (0.10, 0.08, 0.09)]
    one = bench.end_to_end_record(scene, reps, [r.wall_s for r in reps], [[r.scene_seconds_per_second] for r in reps], 1)
    assert one["value"] == 240.0 / 0.08 and one["scenes"] == 4 and "value_by_rank" not in one
    assert one["h2d_bytes_per_scene"] == 800 and one["d2h_bytes_per_scene"] == 300
    # four ranks; the slowest rank's wall time per pass is what the job waits for: the third pass is the best one
    walls = [0.20, 0.16, 0.12]
    by_rank = [[1200.0, 2400.0, 2000.0, 1500.0], [1500.0, 3000.0, 2500.0, 2000.0], [2000.0, 2666.7, 2400.0, 2200.0]]
    four = bench.end_to_end_record(scene, reps, walls, by_rank, 4)
    assert four["value"] == 4 * 240.0 / 0.12 and four["scenes"] == 16 and four["ranks_concurrent"] == 4
    assert four["value_by_rank"] == [2000.0, 2666.7, 2400.0, 2200.0]
    assert abs(four["host_GBps_aggregate"]["both"] - 4 * (3200 + 1200) / 0.12 / 1e9) < 1e-12
    assert four["passes"] == [round(4 * 240.0 / w, 1) for w in walls] and "AT THE SAME TIME" in four["note"]


def test_source_hash_ignores_comments_but_not_code(tmp_path, monkeypatch):
    """profiles/pmc_traffic.json is tied to the kernel sources by bench.source_hash(): a comment or white-space edit must leave a
    collected table valid (round 5 re-collected four configs x three PMC passes seven times for such edits), any change of a token
    must not.  (Whether the COMMITTED table matches this tree is reported by bench.py itself -- `roofline.traffic_note` -- and warned
    about here: a kernel edit must not turn the CPU suite red before the profiles have been collected again.)"""
    import json
    import shutil
    import warnings

    import bench

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    here = bench.source_hash()
    if table["source_hash"] != here:
        warnings.warn(f"profiles/pmc_traffic.json was measured on kernel sources {table['source_hash']}, this tree is {here}: "
                      "re-collect (profiles/tools/collect_profiles.sh)")
    copy = tmp_path / "audiblelight_amd" / "csrc"
    shutil.copytree(os.path.join(root, "audiblelight_amd", "csrc"), copy, ignore=shutil.ignore_patterns("*.o", "*.so"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.source_hash() == here
    src = copy / "al_kernels.hip"
    text = src.read_text()
    src.write_text("// a new first line\n" + text.replace("\n", "   \n", 40).replace("{\n", "{  /* why */\n", 5))
    assert bench.source_hash() == here, "a comment / white-space edit changed the hash"
    src.write_text(text.replace("__launch_bounds__(64)", "__launch_bounds__(128)", 1))
    assert bench.source_hash() != here, "a code edit did not change the hash"
    # literals are code: "//" inside a string is not a comment
    assert bench.strip_c_comments('a = "x // y"; // z') == 'a = "x // y";'
    assert bench.strip_c_comments("f(/* in */ 1,\n  2)  // tail") == "f( 1, 2)"


def test_every_event_meets_the_oracle_on_a_sampled_row():
    """bench.oracle_row_samples: one pseudo-random capsule row of every event that is not compared in full -- the unscaled render
    against the float64 oracle's row, plus the A9 level invariant from the device statistics -- on shrunken cfg3 (moving) and cfg5
    (static, folded FX) scenes through the host-emulated kernels; a corrupted row and a corrupted level are both caught; merged
    into `parity`, parity.events counts the events in full AND the sampled ones."""
    import bench
    from audiblelight_amd import _hip, engine, synthetic
    from oracle import synth_oracle as orc
    from tests import hostemu

    r = engine.Renderer(lib=_hip.Library(hostemu.build()), memory=hostemu.NumpyMemory())
    for name, kw in (("cfg3", dict(scale=0.03, E=3, N=4)), ("cfg5", dict(scale=0.02, E=5, C=6))):
        sc = synthetic.make_scene(name, **kw)
        pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
        res = r.render(pl, sc.sources(), sc.irs)
        cols = lambda e: slice(sc.specs[e].emitter0, sc.specs[e].emitter0 + sc.specs[e].n_emitters)    # noqa: E731
        args = (lambda e, c: sc.irs[c, cols(e), :], lambda e: orc.emitter_gains(sc.irs[:, cols(e), :]))
        rec = bench.oracle_row_samples(sc, res, range(1, len(sc.specs)), *args)
        assert rec["ok"] and rec["events"] == len(sc.specs) - 1 and 0 < rec["rel_rms_worst_row"] < 1e-5
        assert rec["row_of_event"] == {str(e): bench.sampled_row(e, sc.n_capsules) for e in range(1, len(sc.specs))}
        par = bench.merge_row_samples({"events": 1, "ok": True}, rec)
        assert par["events"] == len(sc.specs) and par["events_in_full"] == 1 and par["ok"] and par["rows_sampled"] is rec
        # a wrong sample in the sampled row of the LAST event, then a wrong level: both show
        e = len(sc.specs) - 1
        ev = res.plan.events[e]
        off = int(ev["out_off"]) + bench.sampled_row(e, sc.n_capsules) * int(ev["len"])
        res.spatial[off + 100] += 0.5
        bad = bench.oracle_row_samples(sc, res, range(1, len(sc.specs)), *args)
        assert not bad["ok"] and bad["max_abs_over_peak_worst_row"] > 1e-3 and not bench.merge_row_samples({"events": 1, "ok": True}, bad)["ok"]
        res.spatial[off + 100] -= 0.5
        res.event_scale[1] *= 1.01
        assert not bench.oracle_row_samples(sc, res, range(1, len(sc.specs)), *args)["ok"]


def test_planning_falls_back_to_the_full_library(monkeypatch):
    """Advisor r05: every plan_batch / plan_mixdown call without lib= goes through _hip.get_planner().  Where the planner-only library
    is absent (a tree built before it existed; only AUDIBLELIGHT_HIP_LIB pointing at a custom / sanitizer build) the full library --
    which exports the same al_plan_* symbols -- plans; a RuntimeError with the build hint only when neither is there."""
    from audiblelight_amd import _hip
    from tests import hostemu

    full = hostemu.build()                       # a complete C-ABI library (the kernel sources compiled for the host)
    monkeypatch.setattr(_hip, "_planner", None)
    monkeypatch.setattr(_hip, "_default", None)
    monkeypatch.setenv("AUDIBLELIGHT_PLAN_LIB", "/nonexistent/libaudiblelight_plan.so")
    monkeypatch.setenv("AUDIBLELIGHT_HIP_LIB", full)
    lib = _hip.get_planner()
    assert isinstance(lib, _hip.Library) and lib.path == full
    pl = planning.plan_batch([planning.EventSpec(n_samples=5000, n_emitters=1, snr=10.0)], 2, 700, 8000)
    assert pl.n_partitions >= 1 and pl.log2_block >= 10
    monkeypatch.setattr(_hip, "_planner", None)
    monkeypatch.setattr(_hip, "_default", None)
    monkeypatch.setenv("AUDIBLELIGHT_HIP_LIB", "/nonexistent/libaudiblelight_hip.so")
    with pytest.raises(RuntimeError, match="planner library not found"):
        _hip.get_planner()
    monkeypatch.setattr(_hip, "_planner", None)
    monkeypatch.setattr(_hip, "_default", None)
