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
    assert lib.call("al_abi_version") == 1
    assert lib.call("al_twiddle_bytes", 13) == 8 * 8192 and lib.call("al_twiddle_bytes", 9) == -1
    assert lib.call("al_row_stats_partials", 3, 40000) == 4 * 3 * 3
    assert lib.call("al_noise_workspace_floats", 2, 1000) > 0
    # argument validation happens before any launch, so it is checkable without a GPU
    with pytest.raises(_hip.HipError, match="null batch"):
        lib.call("al_render_batch", None, None)
    bad = _hip.AlBatch(log2_block=3, n_capsules=1, hop=128)
    with pytest.raises(_hip.HipError, match=r"log2_block must be in \[10, 14\]"):
        lib.call("al_ir_spectra", ct.byref(bad), None)


def test_struct_layouts_match_the_header():
    assert _hip.EVENT_DTYPE.itemsize == 56 and _hip.STREAM_DTYPE.itemsize == 32
    assert ct.sizeof(_hip.AlBatch) == 6 * 4 + 2 * 8 + 4 * 4 + 6 * 4 + 16 * 8 + 2 * 4   # 16 pointers, then the two zero-block indices
    assert ct.sizeof(_hip.AlMix) == 6 * 4 + 13 * 8
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
    the GPU box) and rank 0 prints ONE JSON line with n_gpus == 2, a gather figure and the per-step roofline fields."""
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
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["gather"]["bytes_per_rank"] > 0 and 0 < out["roofline"]["path_frac"] <= out["roofline"]["frac"]
    assert "HOST EMULATION" in out["data"]
    # a launcher that sets WORLD_SIZE differently from --gpus is an error, not a silent 1-GPU run
    bad = subprocess.run(cmd, env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in bad.stderr


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
