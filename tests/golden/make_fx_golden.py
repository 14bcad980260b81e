#!/usr/bin/env python
"""Golden vectors for the sample-wise FX that the reference implements ITSELF in numpy: Fade, Invert,
Reverse and the four TimeWarp classes, run through the REAL ``audiblelight.augmentation`` classes of /root/reference
(this container only), incl. the wrap-pad contract of ``Augmentation.process`` (augmentation.py:91-130).

Third-party packages that are absent are stubbed as in make_golden.py.  One of them is called by the TimeWarp family:
``librosa.util.frame(x, frame_length=, hop_length=)``, whose documented result for a 1-D input is the strided view of shape
(frame_length, n_frames); that one function is given a numpy stand-in here (``_frame``), so the TimeWarp vectors pin the
reference's own loop over that array (augmentation.py:1672-1790), not librosa.  Python's global ``random`` is seeded before
every TimeWarp call and the seed is stored with the vector.

    python tests/golden/make_fx_golden.py      ->  tests/golden/reference_fx_vectors.npz
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference  # noqa: E402


def _frame(x, *, frame_length, hop_length, axis=-1, **_):
    x = np.asarray(x)
    assert x.ndim == 1 and axis == -1
    n_frames = 1 + (x.shape[-1] - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return x[idx]                                            # (frame_length, n_frames), librosa >= 0.9 layout


def main():
    _import_reference()
    import audiblelight.augmentation as ref_aug

    ref_aug.librosa.util.frame = _frame
    rng = np.random.default_rng(77)
    sr = 8000
    out = {"sr": np.int64(sr)}
    x = rng.standard_normal(4001).astype(np.float32)
    x /= np.abs(x).max()
    out["x"] = x
    short = x[:700].copy()                                   # shorter than one TimeWarp frame / than the fades
    out["x_short"] = short

    shapes = ref_aug.Fade.FADE_SHAPES
    cases = [(a, "linear", 0.1, 0.2) for a in shapes] + [("half_sine", b, 0.15, 0.05) for b in shapes]
    cases += [("exponential", "logarithmic", 0.0, 0.3), ("quarter_sine", "quarter_sine", 2.0, 2.0)]   # no fade-in; fades > clip
    meta = []
    for i, (a, b, la, lb) in enumerate(cases):
        fx = ref_aug.Fade(sample_rate=sr, fade_in_len=la, fade_out_len=lb, fade_in_shape=a, fade_out_shape=b)
        src = short if la > 1 else x
        out[f"fade_{i}"] = fx(src)
        meta.append(f"{a},{b},{la},{lb},{'short' if la > 1 else 'x'}")
    out["fade_cases"] = np.array(meta)
    out["invert"] = ref_aug.Invert(sample_rate=sr)(x)
    out["reverse"] = ref_aug.Reverse(sample_rate=sr)(x)

    tw = []
    for name in ("TimeWarpSilence", "TimeWarpDuplicate", "TimeWarpRemove", "TimeWarpReverse"):
        for j, (fps, prob, src_name) in enumerate([(7.3, 0.4, "x"), (2.0, 0.15, "x"), (5.0, 0.5, "x_short"), (3.0, 1.0, "x")]):
            fx = getattr(ref_aug, name)(sample_rate=sr, fps=fps, prob=prob)
            seed = 1000 + 10 * len(tw) + j
            random.seed(seed)
            out[f"tw_{len(tw)}"] = fx(out[src_name])
            tw.append(f"{name},{fps},{prob},{src_name},{seed}")
    out["tw_cases"] = np.array(tw)
    # G12: Ambience in file mode (ambience.py:170-214).  Decoding is librosa's business (absent): librosa.load is replaced by a
    # function that hands back the array below, so the vectors pin the reference's OWN channel selection, tiling along
    # channels and time, truncation and per-channel peak normalisation.
    import audiblelight.ambience as ref_amb

    clip3 = (rng.standard_normal((3, 1700)) * np.array([[0.3], [0.6], [0.9]])).astype(np.float32)
    out["amb_clip3"] = clip3
    amb_cases = []
    for j, (rows, channels, duration, seed) in enumerate([(1, 4, 0.5, 5), (3, 3, 0.6, 6), (3, 2, 0.33, 7), (3, 4, 0.1, 8)]):
        src = clip3[:rows]
        ref_amb.librosa.load = lambda *a, _src=src, **k: (_src if _src.shape[0] > 1 else _src[0], sr)
        amb = ref_amb.Ambience(channels=channels, duration=duration, alias=f"amb{j}", filepath=__file__, sample_rate=sr)
        random.seed(seed)
        out[f"amb_{j}"] = np.asarray(amb.load_ambience(ignore_cache=True, normalize=True))
        out[f"amb_raw_{j}"] = np.asarray(amb.load_ambience(ignore_cache=True, normalize=False)) if rows != 3 or channels == 3 else np.zeros(0)
        amb_cases.append(f"{rows},{channels},{duration},{seed}")
    out["amb_cases"] = np.array(amb_cases)
    # G13: DCASE-2024 metadata rows (synthesize.py:742-878), the reference's own function on duck-typed events: static and
    # moving events, two events sharing a file (same source index), two classes, an event clipped by the scene end
    import types

    import audiblelight.synthesize as ref_syn

    def emitter(polar):
        return types.SimpleNamespace(coordinates_relative_polar={m: np.array([p]) for m, p in polar.items()})

    def dcase_event(alias, class_id, filename, start, end, polars):
        return types.SimpleNamespace(alias=alias, class_id=class_id, filename=filename, scene_start=start, scene_end=end,
                                     is_moving=len(polars) > 1, emitters=[emitter(p) for p in polars])

    mics = ["mic000", "mic001"]
    spec = [("e0", 3, "phone.wav", 0.5, 2.3, [[30.4, -10.6, 1.234]]),
            ("e1", 3, "phone.wav", 4.0, 5.0, [[-120.0, 5.0, 2.5]]),
            ("e2", 3, "other_phone.wav", 1.2, 3.7, [[10.0, 0.0, 1.0], [50.0, 20.0, 2.0], [90.0, 10.0, 1.5], [45.0, -5.0, 3.0]]),
            ("e3", 7, "speech.wav", 8.6, 11.0, [[170.2, 44.5, 0.504]])]
    events = []
    for alias, cid, fname, t0, t1, polars in spec:
        per_emitter = [{m: [az + 3.0 * k, el - 1.0 * k, d * (1 + 0.1 * k)] for k, m in enumerate(mics)} for az, el, d in polars]
        events.append(dcase_event(alias, cid, fname, t0, t1, per_emitter))
    scene_ns = types.SimpleNamespace(duration=10.0, state=types.SimpleNamespace(microphones={m: None for m in mics}),
                                     get_events=lambda: events)
    frames = ref_syn.generate_dcase2024_metadata(scene_ns)
    for m in mics:
        df = frames[m].reset_index()
        out[f"dcase_{m}"] = df.to_numpy().astype(np.int64)
    out["dcase_columns"] = np.array(list(frames[mics[0]].reset_index().columns))
    out["dcase_spec"] = np.array(json.dumps(spec))
    path = os.path.join(HERE, "reference_fx_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith(("fade_0", "tw_0", "inv"))},
          os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
