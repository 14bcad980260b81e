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
    path = os.path.join(HERE, "reference_fx_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith(("fade_0", "tw_0", "inv"))},
          os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
