"""BASELINE.json configs[2..4] as parity cases (reduced sizes the oracle finishes in seconds):
moving sources (cfg3), a multi-scene batch in one launch sequence (cfg4), 64 capsules + ambience + folded FX (cfg5)."""
import os

import numpy as np
import pytest

from oracle import synth_oracle as orc
from tests.conftest import assert_parity, pcm16, rel_rms

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def gpu():
    from audiblelight_amd import engine

    return engine.Renderer()


def oracle_event(sc, i):
    sp = sc.specs[i]
    h = sc.irs[:, sp.emitter0: sp.emitter0 + sp.n_emitters, :].astype(np.float64)
    clip = sc.clips[i]
    if sc.gain_db is not None:   # cfg5: raw clip -> [Gain, Invert] -> peak normalisation, as the reference chains them
        clip = orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(clip, sc.gain_db[i])))
    return orc.render_event(clip, h, sp.snr, sp.ref_db, sp.is_moving, sp.duration, sc.sr)["spatial"]


def test_cfg3_moving_sources(gpu):
    from audiblelight_amd import plan as planning, synthetic

    sc = synthetic.make_scene("cfg3", scale=0.05, E=4)            # 4 events x 32 waypoints, 32 capsules
    pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
    res = gpu.render(pl, sc.clips, sc.irs)
    res.check_finite()
    assert int(pl.streams["n_j"].sum()) < 32 * 4 * int(pl.events["n_blocks"].max())   # cross-fade windows are sparse
    for i in (0, 3):
        want = oracle_event(sc, i)
        got = res.spatial_audio(i)
        assert_parity(got, want, TOL)
        assert np.mean(np.abs(got)) == pytest.approx(10 ** ((-65 + sc.specs[i].snr) / 20), rel=1e-5)


def test_cfg4_scene_batch_in_one_launch(gpu):
    """Several independent scenes share one launch sequence (batch.merge_jobs): events concatenated, IR columns offset."""
    from audiblelight_amd import batch, synthetic

    scenes = [synthetic.make_scene("cfg4", scene_index=i, scale=0.04, E=5, C=8) for i in range(3)]
    jobs = [batch.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration,
                           sample_rate=sc.sr, name=f"s{i}") for i, sc in enumerate(scenes)]
    outs = batch.render_merged(gpu, jobs)
    for sc, got in zip(scenes, outs):
        n = len(sc.specs)
        want = orc.mix_scene([oracle_event(sc, i) for i in range(n)], list(zip(sc.starts, sc.ends)), sc.duration, sc.sr,
                             keep_padded=False)["scene"]
        assert_parity(got, want, TOL)


def test_cfg5_64ch_ambience_and_folded_fx(gpu):
    """BASELINE configs[4] reduced: built from the PRODUCT's classes -- core.Event(augmentations=[Gain, Invert]),
    amb.Ambience("white"), Scene.generate() -- and compared with the oracle's chain
    mix(render(peak_normalise(invert(gain(raw))))) + ambience.  No FX kernel runs and no clip comes back to the host:
    the scalar is evaluated on the device (al_clip_scales) and folded into the clip spectra."""
    from audiblelight_amd import ambience as amb, augmentation as aug, core, synthetic
    from audiblelight_amd import synthesize as syn

    syn.set_renderer(gpu)
    try:
        sc = synthetic.make_scene("cfg5", scale=0.03, E=6)
        assert sc.n_capsules == 64 and len(sc.gain_db) == 6
        scene = core.Scene(sc.duration, core.StaticIRState({"em64": sc.irs}), sample_rate=sc.sr, ref_db=-65)
        for i, (raw, sp) in enumerate(zip(sc.clips, sc.specs)):
            scene.add_event(core.Event(f"e{i}", raw, sc.sr, snr=sp.snr, scene_start=sc.starts[i],
                                       augmentations=[aug.Gain(sc.sr, gain_db=sc.gain_db[i]), aug.Invert(sc.sr)]))
        scene.add_ambience(amb.Ambience(channels=64, duration=sc.duration, alias="a", noise="white", ref_db=-65, sample_rate=sc.sr))
        got = scene.generate()["em64"]
        assert all(ev.audio is None and getattr(ev, "_last_chain", None) is None for ev in scene.events.values())
        n = len(sc.specs)
        noise = orc.ambience_noise(0, 64, sc.duration, sc.sr)
        want = orc.mix_scene([oracle_event(sc, i) for i in range(n)], list(zip(sc.starts, sc.ends)), sc.duration, sc.sr,
                             ambiences=[(noise, -65)], keep_padded=False)["scene"]
        assert_parity(got, want, TOL)
        # the same through the engine-level hand-over the bench uses (ClipSource with prescale + normalize)
        from audiblelight_amd import plan as planning

        pl = planning.plan_batch(sc.specs, 64, sc.ir_len, sc.sr)
        res = gpu.render(pl, sc.sources(), sc.irs)
        for i in (0, n - 1):
            assert_parity(res.spatial_audio(i), oracle_event(sc, i), TOL)
    finally:
        syn.set_renderer(None)


def test_cfg5_regime_all_rows(gpu):
    """cfg5's own kernel regime at FULL length: B = 8192, P = 24 partitions (4 s RIR) -> k_spectral_mac<12,12,2,KSPLIT> with
    two full partition tiles, K = 24, C = 64 capsules, clip scales folded on the device, the ambience fused into the mixdown
    over the whole 60 s scene.  2 events built from core.Event(augmentations=[Gain, Invert]) + a white Ambience through
    Scene.generate(); the scene and EVERY row of every event against the oracle."""
    from audiblelight_amd import ambience as amb, augmentation as aug, core, plan as planning, synthetic
    from audiblelight_amd import synthesize as syn
    from tests import mac_regimes as mr

    sc = synthetic.make_scene("cfg5", E=2)
    assert sc.n_capsules == 64 and sc.ir_len == 192000 and len(sc.clips[0]) == 192000 and sc.duration == 60.0
    pl = planning.plan_batch(sc.specs, 64, sc.ir_len, sc.sr)
    assert pl.log2_block == 13 and pl.n_partitions == 24 and int(pl.events["n_blocks"].max()) == 24
    batch = gpu.prepare(pl, sc.sources(), sc.irs)
    assert mr.mac_codes(gpu, batch) == (1121202, 0) and mr.is_split(batch)
    want_events = [oracle_event(sc, i) for i in range(2)]
    res = batch.run()
    res.check_finite()
    for i in range(2):
        mr.check_event_rows(res, i, want_events[i])
    del res, batch
    # ... and the regime the planner picks for the FULL batch of 128 such events (tests/test_gpu_full_size.py): B = 16384 through the
    # quad-tile transforms (csrc/al_quad16.h), 12 partitions in the capsule loop's register tile
    pl14 = planning.plan_batch(sc.specs, 64, sc.ir_len, sc.sr, log2_block=14)
    batch = gpu.prepare(pl14, sc.sources(), sc.irs)
    assert pl14.n_partitions == 12 and mr.mac_codes(gpu, batch) == (3121201, 0) and mr.is_split(batch)
    res = batch.run()
    res.check_finite()
    for i in range(2):
        mr.check_event_rows(res, i, want_events[i])
    del res, batch
    syn.set_renderer(gpu)
    try:
        scene = core.Scene(sc.duration, core.StaticIRState({"em64": sc.irs}), sample_rate=sc.sr, ref_db=-65)
        for i, (raw, sp) in enumerate(zip(sc.clips, sc.specs)):
            scene.add_event(core.Event(f"e{i}", raw, sc.sr, snr=sp.snr, scene_start=sc.starts[i],
                                       augmentations=[aug.Gain(sc.sr, gain_db=sc.gain_db[i]), aug.Invert(sc.sr)]))
        scene.add_ambience(amb.Ambience(channels=64, duration=sc.duration, alias="a", noise="white", ref_db=-65, sample_rate=sc.sr))
        got = scene.generate()["em64"]
        assert got.shape == (64, 2880000) and got.dtype == np.float32
        for i, ev in enumerate(scene.events.values()):
            rows = ev.spatial_audio["em64"]
            for c in range(64):
                assert_parity(rows[c], want_events[i][c], TOL, what=(i, c))
        noise = orc.ambience_noise(0, 64, sc.duration, sc.sr)
        want = orc.mix_scene(want_events, list(zip(sc.starts, sc.ends)), sc.duration, sc.sr, ambiences=[(noise, -65)],
                             keep_padded=False)["scene"]
        for c in range(64):
            assert_parity(got[c], want[c], TOL, what=c)
    finally:
        syn.set_renderer(None)


def oracle_scene(sc):
    n = len(sc.specs)
    return orc.mix_scene([oracle_event(sc, i) for i in range(n)], list(zip(sc.starts, sc.ends)), sc.duration, sc.sr,
                         keep_padded=False)["scene"]


def test_batch_driver_writes_what_the_oracle_mixes(gpu, tmp_path):
    """the pipelined multi-scene driver (H2D / render / device-side frame encoding / D2H / WAV writer)
    against the ORACLE's scenes: float32 frames within the parity tolerance, PCM_16 frames (soundfile's default subtype,
    core.py:1840-1847) within one LSB of lrint(oracle * 32768); float64 IRs (cast in the planner stage, one with an odd row length) and float32 IRs;
    skip_existing leaves written scenes alone (benchmark.py:54-55)."""
    from scipy.io import wavfile

    from audiblelight_amd import batch, synthetic

    scenes = [synthetic.make_scene("cfg1", scene_index=i, scale=0.5) for i in range(4)]
    scenes[3].irs = np.ascontiguousarray(scenes[3].irs[:, :, :-3])   # a row length that is not a multiple of 4 (float64 job below)
    jobs = [batch.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs.astype(np.float64) if i % 2 else sc.irs,
                           starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name=f"s{i}")
            for i, sc in enumerate(scenes)]
    want = [oracle_scene(sc) for sc in scenes]
    got = {}
    drv = batch.BatchDriver(gpu)
    rep = drv.run(jobs, output_dir=str(tmp_path / "f32"), on_scene=got.__setitem__, subtype="FLOAT")
    assert rep.n_scenes == 4 and len(rep.files) == 4 and rep.scene_seconds == pytest.approx(4 * scenes[0].duration)
    for i, sc in enumerate(scenes):
        assert_parity(got[f"s{i}"], want[i], TOL)
        sr, wav = wavfile.read(str(tmp_path / "f32" / f"s{i}.wav"))
        assert sr == sc.sr and wav.dtype == np.float32 and wav.shape == want[i].T.shape
        np.testing.assert_array_equal(wav.T, got[f"s{i}"])          # the file holds exactly the rendered scene
    rep16 = drv.run(jobs, output_dir=str(tmp_path / "pcm"))           # default subtype
    assert rep16.d2h_bytes * 2 == rep.d2h_bytes - sum(w.size * 4 for w in want)   # half the payload, no float copy
    for i, sc in enumerate(scenes):
        sr, wav = wavfile.read(str(tmp_path / "pcm" / f"s{i}.wav"))
        assert wav.dtype == np.int16 and wav.shape == want[i].T.shape
        ref = np.clip(np.rint(want[i].T.astype(np.float64) * 32768.0), -32768, 32767)
        assert np.abs(wav.astype(np.float64) - ref).max() <= 1
        # bit-exact against the encoder's definition applied to the float32 scene the device held
        np.testing.assert_array_equal(wav, pcm16(got[f"s{i}"].T))
    again = drv.run(jobs, output_dir=str(tmp_path / "pcm"), skip_existing=True)
    assert again.n_scenes == 0 and sorted(again.skipped) == [f"s{i}" for i in range(4)]


def test_batch_driver_reports_writer_failures(gpu, tmp_path, monkeypatch):
    """A non-finite scene or a failing callback must surface from run() (the reference raises through
    librosa.util.valid_audio, synthesize.py:398,603), not die silently in the writer thread or hang the producer."""
    from audiblelight_amd import batch, synthetic

    scenes = [synthetic.make_scene("cfg1", scene_index=i, scale=0.25) for i in range(8)]
    scenes[2].clips[1][100] = np.nan
    jobs = [batch.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration,
                           sample_rate=sc.sr, name=f"s{i}") for i, sc in enumerate(scenes)]
    with pytest.raises(ValueError, match="not finite"):
        batch.BatchDriver(gpu, depth=1).run(jobs, output_dir=str(tmp_path))

    def boom(name, arr):
        raise RuntimeError("callback failed")

    with pytest.raises(RuntimeError, match="callback failed"):
        batch.BatchDriver(gpu, depth=1).run(jobs[3:], on_scene=boom)

    # failures in the feeder threads: the scene source raising half way (planner thread), IRs the uploader cannot take
    def source():
        yield from jobs[3:5]
        raise KeyError("scene factory failed")

    seen = []
    with pytest.raises(KeyError, match="scene factory failed"):
        batch.BatchDriver(gpu).run(source(), on_scene=lambda n, a: seen.append(n))
    assert seen == ["s3", "s4"]                                  # what was handed over before the failure still came out
    drv, real, calls = batch.BatchDriver(gpu), gpu.upload_irs, []

    def flaky(irs, **kw):
        calls.append(1)
        if len(calls) == 2:
            raise MemoryError("upload failed")
        return real(irs, **kw)

    monkeypatch.setattr(gpu, "upload_irs", flaky)
    with pytest.raises(MemoryError, match="upload failed"):
        drv.run(jobs[3:], on_scene=lambda n, a: None)


def test_batch_driver_slow_writer_keeps_its_buffer(gpu, tmp_path, monkeypatch):
    """Writers finish out of order: while one of four writer threads sits in a slow ``wavfile.write`` the driver renders a
    dozen more scenes.  Its page-locked frame buffer must not be handed to a later scene's D2H copy before it is done
    (buffers come from a free list, batch.BatchDriver.run): the frames it writes AFTER the stall must still be its own."""
    import time

    from scipy.io import wavfile

    from audiblelight_amd import batch, synthetic

    scenes = [synthetic.make_scene("cfg1", scene_index=i, scale=0.25) for i in range(14)]
    jobs = [batch.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration,
                           sample_rate=sc.sr, name=f"s{i}") for i, sc in enumerate(scenes)]
    real_write, torn = wavfile.write, []

    def slow_write(path, rate, data):
        if path.endswith("s0.wav") or path.endswith("s5.wav"):
            before = np.array(data, copy=True)
            time.sleep(0.8)                       # twelve other scenes pass through the other writers meanwhile
            if not np.array_equal(before, data):
                torn.append(path)
        return real_write(path, rate, data)

    monkeypatch.setattr(wavfile, "write", slow_write)
    got = {}
    rep = batch.BatchDriver(gpu, depth=2, writers=4).run(jobs, output_dir=str(tmp_path), on_scene=got.__setitem__, subtype="FLOAT")
    assert rep.n_scenes == 14 and not torn
    for i in range(14):
        _, wav = wavfile.read(str(tmp_path / f"s{i}.wav"))
        np.testing.assert_array_equal(wav.T, got[f"s{i}"])
    assert_parity(got["s0"], oracle_scene(scenes[0]), TOL)
    assert_parity(got["s13"], oracle_scene(scenes[13]), TOL)


def test_render_dataset_layout_and_audio(gpu, tmp_path):
    """The reference's dataset loop (scripts/generate/benchmark.py:35-82, generate_with_random_events.py:222-238) on the
    pipelined driver: Scene objects in, ``<out>/<scene>/audio_out_<mic>.wav`` + ``metadata_out.json`` with "time" out;
    audio checked against the oracle (two microphones with different capsule counts, an FX chain, an ambience);
    a second pass skips what exists without building the scenes."""
    import json

    from scipy.io import wavfile

    from audiblelight_amd import ambience as amb, augmentation as aug, batch, core
    from audiblelight_amd import synthesize as syn

    syn.set_renderer(gpu)
    try:
        sr, dur = 16000, 1.5
        built = []

        def make(i):
            rng = np.random.default_rng(900 + i)
            irs = {"mic_a": (rng.standard_normal((4, 3, 900)) * np.exp(-np.arange(900) / 200.0)).astype(np.float32),
                   "mic_b": (rng.standard_normal((2, 3, 900)) * np.exp(-np.arange(900) / 150.0)).astype(np.float64)}
            scene = core.Scene(dur, core.StaticIRState(irs), sample_rate=sr, ref_db=-60)
            raws = [rng.standard_normal(n).astype(np.float32) for n in (9000, 12000, 7000)]
            fx = [[], [aug.Gain(sr, gain_db=-3.0), aug.Invert(sr)], [aug.Fade(sr, 0.02, 0.03, "linear", "linear")]]
            for k, raw in enumerate(raws):
                scene.add_event(core.Event(f"e{k}", raw, sr, snr=8.0 + 3 * k, scene_start=0.1 + 0.3 * k, augmentations=fx[k]))
            built.append(i)
            return scene, raws, irs

        made = {}

        def factory(i):
            def build():
                made[i] = make(i)
                return made[i][0]
            return build

        rep = batch.render_dataset(((f"scene_{i:03d}", factory(i)) for i in range(3)), str(tmp_path), subtype="FLOAT")
        assert rep.n_scenes == 6 and built == [0, 1, 2]
        for i in range(3):
            scene, raws, irs = made[i]
            clips = [orc.peak_normalise_clip(raws[0]), orc.peak_normalise_clip(orc.fx_invert(orc.fx_gain(raws[1], -3.0))),
                     orc.peak_normalise_clip(orc.fx_fade(raws[2], sr, 0.02, 0.03, "linear", "linear"))]
            folder = tmp_path / f"scene_{i:03d}"
            meta = json.load(open(folder / "metadata_out.json"))
            assert meta["time"] > 0 and list(meta["events"]) == ["e0", "e1", "e2"]
            for mic, h in irs.items():
                spat = [orc.render_event(c, h[:, [k], :].astype(np.float64), 8.0 + 3 * k, ref_db=-60, sr=sr)["spatial"]
                        for k, c in enumerate(clips)]
                slots = [(e.scene_start, e.scene_end) for e in scene.events.values()]
                want = orc.mix_scene(spat, slots, dur, sr, keep_padded=False)["scene"]
                rate, wav = wavfile.read(str(folder / f"audio_out_{mic}.wav"))
                assert rate == sr and wav.shape == want.T.shape
                assert_parity(wav.T, want, TOL)
        again = batch.render_dataset(((f"scene_{i:03d}", factory(i)) for i in range(4)), str(tmp_path), subtype="FLOAT")
        assert again.n_scenes == 2 and built == [0, 1, 2, 3] and sorted(again.skipped) == ["scene_000", "scene_001", "scene_002"]
        # a folder an interrupted run left WITHOUT its metadata file (written last) is not "done": it is rendered again
        os.remove(tmp_path / "scene_001" / "metadata_out.json")
        os.remove(tmp_path / "scene_001" / "audio_out_mic_b.wav")
        redo = batch.render_dataset(((f"scene_{i:03d}", factory(i)) for i in range(4)), str(tmp_path), subtype="FLOAT")
        assert redo.n_scenes == 2 and built == [0, 1, 2, 3, 1] and sorted(redo.skipped) == ["scene_000", "scene_002", "scene_003"]
        assert (tmp_path / "scene_001" / "metadata_out.json").exists() and (tmp_path / "scene_001" / "audio_out_mic_b.wav").exists()
        # one process per GPU, same stream of scenes: rank r renders every world_size-th scene into the shared folder and
        # never builds the others (two "ranks" run one after the other here; there is no collective to wait for)
        del built[:]
        shared = tmp_path / "sharded"
        stream = lambda: ((f"scene_{i:03d}", factory(10 + i)) for i in range(5))  # noqa: E731
        r0 = batch.render_dataset(stream(), str(shared), subtype="FLOAT", rank=0, world_size=2)
        assert built == [10, 12, 14] and r0.n_scenes == 6
        r1 = batch.render_dataset(stream(), str(shared), subtype="FLOAT", rank=1, world_size=2)
        assert built == [10, 12, 14, 11, 13] and r1.n_scenes == 4
        assert sorted(os.listdir(shared)) == [f"scene_{i:03d}" for i in range(5)]
    finally:
        syn.set_renderer(None)


def test_hip_graph_replay_matches_eager(gpu):
    """cfg1-sized scene recorded into a HIP graph (engine.CapturedScene): replays equal the eager launches bit for bit,
    also after new clips are written into the captured buffers."""
    import torch

    from audiblelight_amd import engine, plan as planning, synthetic

    sc = synthetic.make_scene("cfg1")
    pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
    n = len(sc.clips)
    mp = planning.plan_mixdown(sc.starts, sc.ends, [len(c) for c in sc.clips], [sc.n_capsules] * n, pl.events["out_off"],
                               list(range(n)), sc.duration, sc.sr, sc.n_capsules)
    batch = gpu.prepare(pl, sc.clips, sc.irs)
    mix = gpu.prepare_mixdown(mp, batch.result(), [])
    batch.run()
    eager = gpu.mem.download(mix.run()).copy()
    cap = engine.CapturedScene(batch, mix)
    for _ in range(2):
        _, scene = cap.replay()
        np.testing.assert_array_equal(gpu.mem.download(scene), eager)
    # new input in the same buffers: negate every clip -> the scene is negated exactly
    batch.bufs["audio"].neg_()
    _, scene = cap.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(gpu.mem.download(scene), -eager)
    want = orc.mix_scene([oracle_event(sc, i) for i in range(n)], list(zip(sc.starts, sc.ends)), sc.duration, sc.sr,
                         keep_padded=False)["scene"]
    assert_parity(-gpu.mem.download(scene)[: want.size].reshape(want.shape), want, TOL)


def test_bench_collectives_on_rccl_with_one_rank():
    """bench.py's multi-rank plumbing on the REAL backend (RCCL init with device_id, barrier, MAX all-reduce of the step
    time, the end-of-job gather) with the one GPU a test box has; the 2-rank logic itself is covered by the gloo test
    (tests/test_host_logic.py::test_bench_gpus_flag_spawns_that_many_ranks)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["AL_BENCH_FORCE_DIST"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "cfg1", "--steps", "20", "--cpu-events", "0",
                          "--cpu-workers", "0", "--end-to-end", "0", "--dropin", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["gather"]["backend"] == "nccl" and out["gather"]["bytes_total"] == 4 * 4 * 240000
    assert out["data"] == "synthetic" and out["value"] > 0 and out["gather"]["ranks_seen"] == [0]
    assert out["timing"]["repeats"] == 3 and out["steps"] == 20
    assert out["gather"]["xgmi_link_peak_GBps"] == 153.0 and out["gather"]["render_ms"] == out["ms_per_step"]
    # the PCIe-inclusive legs as a multi-rank run makes them: behind RCCL barriers, pass times reduced with MAX over the ranks
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "cfg1", "--steps", "5", "--repeats", "1", "--cpu-events", "0",
                          "--cpu-workers", "0", "--end-to-end", "4", "--dropin", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    legs = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert legs["end_to_end"]["value"] > 0 and legs["end_to_end"]["scenes"] == 4 and len(legs["end_to_end"]["passes"]) == 3
    assert legs["end_to_end_dropin"]["value"] > 0 and legs["end_to_end_dropin"]["scenes"] == 2
    # the other two modes on the same backend: a batch of scenes all gathered, one scene with its capsules "sharded" over one rank
    common = [sys.executable, os.path.join(root, "bench.py"), "--config", "cfg1", "--steps", "5", "--repeats", "1", "--cpu-events", "0",
              "--cpu-workers", "0"]
    for extra, check in ((["--total-scenes", "3"], lambda o: o["config"]["total_scenes"] == 3 and o["gather"]["bytes_total"] == 3 * 4 * 4 * 240000
                          and o["gather"]["overlapped"]["bit_exact"] and o["gather"]["overlapped"]["step_with_gather_ms"] > 0
                          and o["host_share"]["threads_by_rank"] == [o["host_share"]["threads"]] and not o["host_share"]["pinned"]),
                         (["--shard", "capsules"], lambda o: o["config"]["capsules_this_rank"] == 4 and o["gather"]["within_tolerance"])):
        res = subprocess.run(common + extra, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
        assert out["gather"]["backend"] == "nccl" and out["scaling"] == "strong" and check(out), out


def test_float64_irs_host_cast_equals_device_cast(gpu, monkeypatch):
    """What the reference's WorldState.get_irs() returns is float64: the thread-pool cast into page-locked memory (default
    for synchronous renders) and the device-side cast (batch driver) put the same float32 bits into HBM, odd row lengths
    included (re-pitched on the device either way)."""
    import torch

    rng = np.random.default_rng(3)
    for shape in ((3, 4, 20_001), (2, 3, 400_000)):
        irs = rng.standard_normal(shape)
        a, sa = gpu.upload_irs(irs, host_cast=True)
        b, sb = gpu.upload_irs(irs, host_cast=False)
        assert sa == sb and torch.equal(a, b)
        pitch = sa[1]
        got = gpu.mem.download(a)[: shape[0] * shape[1] * pitch].reshape(shape[0], shape[1], pitch)
        np.testing.assert_array_equal(got[:, :, : shape[2]], irs.astype(np.float32))


def test_render_dataset_writes_dcase_csvs(gpu, tmp_path):
    """The dataset loop with metadata_dcase=True on the reference-format scene (class indices + emitter positions in its
    metadata): audio, JSON and one DCASE-2024 CSV per microphone in the scene's folder."""
    import json

    import pandas as pd

    from audiblelight_amd import batch, core, synthesize as syn

    here = os.path.join(os.path.dirname(__file__), "golden")
    arrays = np.load(os.path.join(here, "reference_scene_arrays.npz"))
    meta = json.load(open(os.path.join(here, "reference_scene.json")))

    def scene():
        return core.Scene.from_dict(meta, {a: arrays[f"clip_{a}"] for a in meta["events"]},
                                    {m: arrays[f"irs_{m}"] for m in meta["state"]["microphones"]})

    syn.set_renderer(gpu)
    try:
        rep = batch.render_dataset([("ref", scene)], str(tmp_path), metadata_dcase=True, subtype="FLOAT")
        assert rep.n_scenes == len(meta["state"]["microphones"])
        want = syn.generate_dcase2024_metadata(scene())
        for mic in meta["state"]["microphones"]:
            assert (tmp_path / "ref" / f"audio_out_{mic}.wav").exists() and (tmp_path / "ref" / "metadata_out.json").exists()
            df = pd.read_csv(tmp_path / "ref" / f"metadata_out_{mic}.csv", header=None)
            np.testing.assert_array_equal(df.to_numpy(), want[mic].reset_index().to_numpy())
    finally:
        syn.set_renderer(None)
