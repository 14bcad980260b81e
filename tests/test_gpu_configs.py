"""BASELINE.json configs[2..4] as parity cases (reduced sizes the oracle finishes in seconds):
moving sources (cfg3), a multi-scene batch in one launch sequence (cfg4), 64 capsules + ambience + folded FX (cfg5)."""
import numpy as np
import pytest

from oracle import synth_oracle as orc
from tests.conftest import rel_rms

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def gpu():
    from audiblelight_amd import engine

    return engine.Renderer()


def oracle_event(sc, i):
    sp = sc.specs[i]
    h = sc.irs[:, sp.emitter0: sp.emitter0 + sp.n_emitters, :].astype(np.float64)
    return orc.render_event(sc.clips[i] * np.float32(sp.gain), h, sp.snr, sp.ref_db, sp.is_moving, sp.duration, sc.sr)["spatial"]


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
        assert rel_rms(got, want) < TOL
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
        assert rel_rms(got, want) < TOL


def test_cfg5_64ch_ambience_and_folded_fx(gpu):
    from audiblelight_amd import ambience as amb, plan as planning, synthetic
    from audiblelight_amd import synthesize as syn

    syn.set_renderer(gpu)
    try:
        sc = synthetic.make_scene("cfg5", scale=0.03, E=6)
        assert sc.n_capsules == 64 and all(sp.gain < 0 for sp in sc.specs)    # Invert folded into the gain
        pl = planning.plan_batch(sc.specs, 64, sc.ir_len, sc.sr)
        res = gpu.render(pl, sc.clips, sc.irs)
        n = len(sc.specs)
        mix = planning.plan_mixdown(sc.starts, sc.ends, [len(c) for c in sc.clips], [64] * n, pl.events["out_off"],
                                    list(range(n)), sc.duration, sc.sr, 64)
        a = amb.Ambience(channels=64, duration=sc.duration, alias="a", noise="white", ref_db=-65, sample_rate=sc.sr)
        dev = syn._ambience_on_device(gpu, a, (64, mix.n_samples))
        got = gpu.mem.download(gpu.mixdown(mix, res, [dev]))[: 64 * mix.n_samples].reshape(64, -1)
        noise = orc.ambience_noise(0, 64, sc.duration, sc.sr)
        want = orc.mix_scene([oracle_event(sc, i) for i in range(n)], list(zip(sc.starts, sc.ends)), sc.duration, sc.sr,
                             ambiences=[(noise, -65)], keep_padded=False)["scene"]
        assert rel_rms(got, want) < TOL
    finally:
        syn.set_renderer(None)


def test_batch_driver_matches_direct_render(gpu, tmp_path):
    """pipelined multi-scene driver (async H2D / render / D2H + WAV writer) gives the same audio
    as rendering each scene on its own; float64 IRs go through the device-side ingest kernel."""
    from scipy.io import wavfile

    from audiblelight_amd import batch, plan as planning, synthetic

    scenes = [synthetic.make_scene("cfg1", scene_index=i, scale=0.5) for i in range(4)]
    jobs = [batch.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs.astype(np.float64) if i % 2 else sc.irs,
                           starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name=f"s{i}")
            for i, sc in enumerate(scenes)]
    got = {}
    rep = batch.BatchDriver(gpu).run(jobs, output_dir=str(tmp_path), on_scene=lambda name, arr: got.__setitem__(name, arr.copy()))
    assert rep.n_scenes == 4 and len(rep.files) == 4 and rep.scene_seconds == pytest.approx(4 * scenes[0].duration)
    for i, sc in enumerate(scenes):
        pl = planning.plan_batch(sc.specs, sc.n_capsules, sc.ir_len, sc.sr)
        res = gpu.render(pl, sc.clips, sc.irs)
        mix = planning.plan_mixdown(sc.starts, sc.ends, [len(c) for c in sc.clips], [sc.n_capsules] * len(sc.clips),
                                    pl.events["out_off"], list(range(len(sc.clips))), sc.duration, sc.sr, sc.n_capsules)
        want = gpu.mem.download(gpu.mixdown(mix, res))[: sc.n_capsules * mix.n_samples].reshape(sc.n_capsules, -1)
        np.testing.assert_allclose(got[f"s{i}"], want, rtol=1e-6, atol=1e-12)
        sr, wav = wavfile.read(str(tmp_path / f"s{i}.wav"))
        assert sr == sc.sr and wav.shape == (mix.n_samples, sc.n_capsules)
        np.testing.assert_array_equal(wav.T, got[f"s{i}"])


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
    assert rel_rms(-gpu.mem.download(scene)[: want.size].reshape(want.shape), want) < TOL
