#!/usr/bin/env python
"""Differential fuzz against the REAL reference (build container only: /root/reference does not exist on the GPU box).

Random small scenes -- static, moving and zero-emitter events, snr drawn from {0, negative, U(5, 30)}, some clips and IRs all
zeros, events running past the scene end or not reaching it, an optional coloured-noise ambience -- are rendered twice from the
same duck-typed objects: by the reference's own ``render_audio_for_all_scene_events`` + ``generate_scene_audio_from_events``
(numpy / scipy, imported from /root/reference exactly as tests/golden/make_golden.py does) and by this package's functions of
the same names over the host-emulated kernels (the unmodified kernel sources compiled for the CPU; the gfx950 build runs the
same scenes in the GPU suite through the oracle).  Every event's ``spatial_audio`` and every ``scene.audio`` must agree within
the contract's bound (1e-4 relative RMS and max-abs / max|ref|) AND carry the reference's dtype (a float32 clip tiled over the
capsules stays float32, synthesize.py:572-599); silent results must be silent on both sides.

    python tests/golden/differential_fuzz.py [FIRST LAST]        (default 0 200; prints the worst case, exit code 1 on a failure)
"""
import copy
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (the reference import machinery and the duck-typed FakeEvent)


def random_scene(seed, amb_mod):
    rng = np.random.default_rng(70_000 + seed)
    sr = int(rng.choice([8000, 16000]))
    n_caps = int(rng.integers(1, 5))
    dur = float(rng.uniform(0.6, 1.6))
    lir = int(rng.integers(40, 1500))
    events, irs = {}, []
    for i in range(int(rng.integers(1, 6))):
        kind = rng.choice(["static", "static", "static", "moving", "tiled"])
        n = int(rng.integers(700 if kind == "moving" else 30, 7000))
        ne = {"static": 1, "tiled": 0, "moving": int(rng.integers(2, 6))}[kind]
        a = mg.make_clip(rng, n)
        if rng.random() < 0.12:
            a = np.zeros_like(a)                                    # a silent clip
        h = mg.make_irs(rng, n_caps, ne, lir) if ne else np.zeros((n_caps, 0, lir))
        if ne and rng.random() < 0.12:
            h[:, int(rng.integers(0, ne)), :] = 0.0                  # an all-zero IR (one emitter of a moving event, or the static one)
        snr = float(rng.choice([0.0, -float(rng.uniform(1, 10)), float(rng.uniform(5, 30)), float(rng.uniform(5, 30))]))
        start = float(rng.uniform(-0.05, dur * 1.02))               # some start before 0, some at or beyond the scene's end
        if seed >= 100_000:     # the adversarial range: clips that are not peak-normalised, IRs far from unit scale, slots on half samples
            a = (a * float(rng.choice([1.0, 1.0, 1e-6, 1e-20, 3e5]))).astype(np.float32)
            h = h * float(rng.choice([1.0, 1.0, 1e-8, 1e4]))
            h = h.astype(np.float32).astype(np.float64)
            if rng.random() < 0.5:
                start = (int(rng.integers(0, int(dur * sr))) + 0.5) / sr     # round-half-even on both slot ends (synthesize.py:361-362)
        # (no dry render for snr = 0: the reference scales it by db_to_multiplier(., mean|0|) = 10^(dB/20) / tiny, i.e. to 1e305 in
        # float64 -- its own output is overflow-scale garbage there, and nothing a float32 path could or should reproduce)
        dry = ne == 1 and snr != 0.0 and rng.random() < 0.25
        events[f"ev{i}"] = mg.FakeEvent(f"ev{i}", a, ne, snr, sr, scene_start=start, is_moving=ne > 1,
                                        ref_ir_channel=int(rng.integers(0, n_caps)) if dry else None,
                                        direct_path_time_ms=[2, 25] if dry else None)
        irs.append(h)
    mic_ir = np.concatenate(irs, axis=1)
    if mic_ir.shape[1] == 0:                                        # the reference refuses a WorldState without emitters
        return None
    # dtypes the reference's arithmetic keeps or widens (a generator of its own: the scenes of earlier records stay the same):
    # float32 IRs (static renders and dry renders stay float32), the odd float64 clip (a tiled clip stays float64)
    drng = np.random.default_rng(90_000 + seed)
    if drng.random() < 0.3:
        mic_ir = mic_ir.astype(np.float32)
    for ev in events.values():
        if drng.random() < 0.15:
            ev._audio = ev._audio.astype(np.float64)
    noise = rng.choice([None, None, "white", "pink", 1.5])
    ambience = {}
    if noise is not None:
        ambience["a0"] = amb_mod.Ambience(channels=n_caps, duration=dur, alias="a0", noise=noise if isinstance(noise, str) else float(noise),
                                          ref_db=float(-rng.uniform(40, 70)), sample_rate=sr)
    state = types.SimpleNamespace(irs={"mic000": mic_ir}, get_irs=lambda: {"mic000": mic_ir}, simulate=lambda: None,
                                  microphones={"mic000": object()}, num_emitters=mic_ir.shape[1], name="fake")
    return types.SimpleNamespace(state=state, events=events, ambience=ambience, audio={}, ref_db=float(-rng.uniform(50, 70)), duration=dur,
                                 sample_rate=sr)


def fx_and_noise_cases(seed, ref_aug, ref_amb, our_aug, our_amb):
    """[(label, ours, reference's)] for one seed: the sample-wise FX the reference implements itself in numpy (Fade with random
    shapes and lengths incl. fades longer than the clip, Invert, Reverse, the four TimeWarp classes under the same Python random
    seed) and powerlaw_psd_gaussian for a random exponent / length / fmin / seed (augmentation.py:1403-1790, ambience.py:271-375)."""
    import random

    rng = np.random.default_rng(80_000 + seed)
    sr = int(rng.choice([8000, 16000, 22050]))
    x = mg.make_clip(rng, int(rng.integers(300, 9000)))
    out = []
    shapes = list(ref_aug.Fade.FADE_SHAPES)
    a, b = rng.choice(shapes), rng.choice(shapes)
    la, lb = float(rng.choice([0.0, rng.uniform(0.01, 0.4), 3.0])), float(rng.choice([0.0, rng.uniform(0.01, 0.4), 3.0]))
    kw = dict(sample_rate=sr, fade_in_len=la, fade_out_len=lb, fade_in_shape=str(a), fade_out_shape=str(b))
    out.append((f"Fade{a, b, round(la, 3), round(lb, 3)}", our_aug.Fade(**kw)(x), ref_aug.Fade(**kw)(x)))
    out.append(("Invert", our_aug.Invert(sample_rate=sr)(x), ref_aug.Invert(sample_rate=sr)(x)))
    out.append(("Reverse", our_aug.Reverse(sample_rate=sr)(x), ref_aug.Reverse(sample_rate=sr)(x)))
    name = str(rng.choice(["TimeWarpSilence", "TimeWarpDuplicate", "TimeWarpRemove", "TimeWarpReverse"]))
    fps, prob = float(rng.uniform(1.5, 12.0)), float(rng.uniform(0.05, 1.0))
    random.seed(seed)
    theirs = getattr(ref_aug, name)(sample_rate=sr, fps=fps, prob=prob)(x)
    random.seed(seed)
    mine = getattr(our_aug, name)(sample_rate=sr, fps=fps, prob=prob)(x)
    out.append((f"{name}(fps={fps:.2f}, prob={prob:.2f})", mine, theirs))
    beta = float(rng.choice([0.0, 1.0, 2.0, -1.0, rng.uniform(-2, 3)]))
    n = int(rng.choice([rng.integers(8, 5000), 4096, 1000, 2 * 3 * 5 * 7 * 11, 4001]))
    fmin = float(rng.choice([0.0, 0.0, rng.uniform(0.0, 0.5)]))
    shape = (int(rng.integers(1, 4)), n)
    sd = int(rng.integers(0, 10_000))
    out.append((f"powerlaw(beta={beta:.2f}, n={n}, fmin={fmin:.3f})", our_amb.powerlaw_psd_gaussian(beta, shape, fmin=fmin, seed=sd),
                ref_amb.powerlaw_psd_gaussian(beta, shape, fmin=fmin, seed=sd)))
    return out


SMOOTH = sorted({2 ** a * 3 ** b * 5 ** c * 7 ** d for a in range(11) for b in range(4) for c in range(3) for d in range(2)
                 if 32 <= 2 ** a * 3 ** b * 5 ** c * 7 ** d <= 1024})


def geometry_case(seed, ref_syn, ours):
    """One moving event under a random STFT geometry the reference accepts (win >= hop, fft <= 2*hop + win; ANY fft size -- half the
    seeds draw a product of 2, 3, 5, 7, the other half any integer, primes included: Bluestein), rendered by the reference's
    render_event_audio and by ours (synthesize.py:507-608,277-310)."""
    rng = np.random.default_rng(85_000 + seed)
    hop = int(rng.integers(16, 200))
    win = int(rng.integers(hop, 3 * hop + 1))
    ok = [f for f in SMOOTH if f <= 2 * hop + win]
    fft = int(rng.choice(ok[-10:])) if rng.random() < 0.7 else int(rng.choice(ok))
    if seed % 2:                                                    # any size at all (a generator of its own: the other draws stay)
        fft = int(np.random.default_rng(86_000 + seed).integers(max(32, (2 * hop + win) // 3), min(1024, 2 * hop + win) + 1))
    n_irs, n_caps, sr = int(rng.integers(2, 6)), int(rng.integers(1, 4)), 8000
    a, h = mg.make_clip(rng, int(rng.integers(1500, 6000))), mg.make_irs(rng, n_caps, n_irs, int(rng.integers(100, 1200)))
    snr = float(rng.uniform(5, 30))
    ev_a = mg.FakeEvent("g", a, n_irs, snr, sr, is_moving=True)
    ev_b = mg.FakeEvent("g", a, n_irs, snr, sr, is_moving=True)
    ref_syn.render_event_audio(ev_a, h, "mic000", ref_db=-60, fft_size=fft, win_size=win, hop_size=hop)
    ours.render_event_audio(ev_b, h, "mic000", ref_db=-60, fft_size=fft, win_size=win, hop_size=hop)
    return f"moving event, fft/win/hop = {fft}/{win}/{hop}", ev_b.spatial_audio["mic000"], ev_a.spatial_audio["mic000"]


def errors(got, ref):
    if np.asarray(got).dtype != np.asarray(ref).dtype:      # a drop-in hands back the reference's dtype, not only its values
        return float("inf"), float("inf")
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if got.shape != ref.shape:
        return float("inf"), float("inf")
    if not np.isfinite(got).all():
        return float("inf"), float("inf")
    peak, rms = np.abs(ref).max() if ref.size else 0.0, np.sqrt(np.mean(ref ** 2)) if ref.size else 0.0
    if peak == 0:
        worst = float(np.abs(got).max()) if got.size else 0.0
        return worst, worst
    return float(np.sqrt(np.mean((got - ref) ** 2)) / rms), float(np.abs(got - ref).max() / peak)


def main():
    first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
    from audiblelight_amd import _hip, engine, synthesize as ours
    from tests import hostemu                       # this repo's tests package, before the reference tree goes onto sys.path

    ref_syn, ref_amb = mg._import_reference()

    ours.set_renderer(engine.Renderer(lib=_hip.Library(hostemu.build()), memory=hostemu.NumpyMemory()))
    worst, bad, n_cmp, n_silent = (0.0, None), [], 0, 0
    for seed in range(first, last):
        a = random_scene(seed, ref_amb)
        if a is None:
            continue
        b = copy.deepcopy(a)
        b.state.get_irs = lambda irs=b.state.irs: irs
        with np.errstate(all="ignore"):
            ref_syn.render_audio_for_all_scene_events(a)
            ref_syn.generate_scene_audio_from_events(a)
        ours.render_audio_for_all_scene_events(b)
        ours.generate_scene_audio_from_events(b)
        pairs = [("scene", b.audio["mic000"], a.audio["mic000"])]
        for k in a.events:
            pairs.append((k, b.events[k].spatial_audio["mic000"], a.events[k].spatial_audio["mic000"]))
            if a.events[k].ref_ir_channel is not None:
                pairs.append((k + ".dry", b.events[k]._spatial_audio_dry["mic000"], a.events[k]._spatial_audio_dry["mic000"]))
            if "mic000" in a.events[k]._spatial_audio_padded:
                pairs.append((k + ".padded", b.events[k]._spatial_audio_padded["mic000"], a.events[k]._spatial_audio_padded["mic000"]))
        for what, got, ref in pairs:
            rms, mx = errors(got, ref)
            n_cmp += 1
            n_silent += float(np.abs(np.asarray(ref)).max() if np.asarray(ref).size else 0.0) == 0.0
            if max(rms, mx) > worst[0]:
                worst = (max(rms, mx), (seed, what))
            if rms > 1e-4 or mx > 1e-4:
                bad.append((seed, what, rms, mx, str(np.asarray(got).dtype), str(np.asarray(ref).dtype)))
    # the FX the reference implements in numpy and its coloured-noise generator, same seeds on both sides
    import audiblelight.augmentation as ref_aug
    from audiblelight_amd import ambience as our_amb, augmentation as our_aug

    sys.path.insert(0, HERE)
    from make_fx_golden import _frame

    ref_aug.librosa.util.frame = _frame           # librosa is absent: its documented framing, as in make_fx_golden.py
    n_fx = 0
    for seed in range(first, last):
        for what, got, ref in fx_and_noise_cases(seed, ref_aug, ref_amb, our_aug, our_amb) + [geometry_case(seed, ref_syn, ours)]:
            rms, mx = errors(got, ref)
            n_cmp, n_fx = n_cmp + 1, n_fx + 1
            if max(rms, mx) > worst[0]:
                worst = (max(rms, mx), (seed, what))
            if rms > 1e-4 or mx > 1e-4:
                bad.append((seed, what, rms, mx, str(np.asarray(got).dtype), str(np.asarray(ref).dtype)))
    ours.set_renderer(None)
    print(f"{n_fx} of the arrays are FX / coloured-noise outputs (Fade, Invert, Reverse, TimeWarp*, powerlaw_psd_gaussian) and moving events "
          f"under random STFT geometries")
    print(f"seeds {first}..{last - 1}: {n_cmp} arrays compared with the reference's own ({n_silent} of them silent), {len(bad)} outside 1e-4; "
          f"worst error {worst[0]:.2e} at {worst[1]}")
    for item in bad[:20]:
        print("FAILED", item)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
