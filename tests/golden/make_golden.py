#!/usr/bin/env python
"""Generate golden vectors by running the REAL reference (this container only).

Imports ``/root/reference/audiblelight`` with ``sys.modules`` stand-ins for the third-party
packages that are absent here (none of them does arithmetic on the hot path), drives the
reference's own functions with duck-typed Scene/Event objects on seeded synthetic inputs
and writes inputs + outputs as ``.npz`` next to this script.  Nothing from the reference
(source or bytecode) is stored: only arrays.

    python tests/golden/make_golden.py

The GPU box never runs this script (``/root/reference`` does not exist there).
"""
import importlib.metadata
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    for name in ["librosa", "librosa.util", "librosa.effects", "soundfile", "trimesh", "trimesh.visual",
                 "loguru", "deepdiff", "pedalboard", "pysofaconventions", "rlr_audio_propagation", "rtree",
                 "pyroomacoustics", "gdown", "h5py", "cv2", "pyvista", "netCDF4", "vtk"]:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = MagicMock()
    real_version = importlib.metadata.version
    importlib.metadata.version = lambda n: "0.1.2" if n == "audiblelight" else real_version(n)
    sys.path.insert(0, REF)
    import audiblelight.ambience as amb
    import audiblelight.synthesize as syn

    def valid_audio(y, **_):  # what librosa.util.valid_audio checks
        if not isinstance(y, np.ndarray) or not np.issubdtype(y.dtype, np.floating):
            raise ValueError("not float ndarray")
        if not np.isfinite(y).all():
            raise ValueError("not finite")
        return True

    syn.librosa.util.valid_audio = valid_audio
    amb.librosa.util.valid_audio = valid_audio
    return syn, amb


class FakeEvent:
    """Exposes exactly the attribute list of SURVEY.md §8a row A15."""

    def __init__(self, alias, audio, n_emitters, snr, sr, scene_start=0.0, is_moving=False,
                 ref_ir_channel=None, direct_path_time_ms=None):
        self.alias = alias
        self._audio = audio
        self.n = n_emitters
        self.snr = snr
        self.sample_rate = sr
        self.duration = len(audio) / sr
        self.scene_start = scene_start
        self.scene_end = scene_start + self.duration
        self.is_moving = is_moving
        self.ref_ir_channel = ref_ir_channel
        self.direct_path_time_ms = direct_path_time_ms
        self.spatial_audio = {}
        self._spatial_audio_padded = {}
        self._spatial_audio_dry = {}
        self._spatial_audio_dry_padded = {}

    def __len__(self):
        return self.n

    def load_audio(self, ignore_cache=False, normalize=True):
        return self._audio


def make_clip(rng, n):
    a = rng.standard_normal(n).astype(np.float32)
    return a / np.max(np.abs(a) + np.finfo(np.float32).tiny)


def make_irs(rng, c, n, l):
    t = np.arange(l)
    h = rng.standard_normal((c, n, l)) * np.exp(-t / (l / 6.9))
    for ci in range(c):
        for ni in range(n):
            h[ci, ni, rng.integers(4, 40)] += 1.0
    # values exactly representable in float32 (what the HIP path ingests); reference sees float64
    return h.astype(np.float32).astype(np.float64)


def main():
    syn, amb = _import_reference()
    out = {}
    rng = np.random.default_rng(20260529)
    sr = 8000

    # G9 scalar known answers (reference tests/test_synthesize.py:42-57,307-337)
    xs = [np.array([0.0, 0.5, -0.5, 1.0, -1.0]), np.array([0.0, 0.25, -0.25]), np.array([1.0, 2.0, 3.0]),
          np.array([-1e-10, 1e-10]), np.zeros(5)]
    snrs = [2.0, 1.0, 6.0, 0.5, 3.0]
    for i, (x, s) in enumerate(zip(xs, snrs)):
        out[f"g9_snr_in{i}"] = x
        out[f"g9_snr_out{i}"] = syn.apply_snr(x, s)
    out["g9_snr_vals"] = np.array(snrs)
    dbs = np.array([0, 6.0206, -6.0206, 20.0, -20.0])
    lv = np.array([1.0, 1.0, 1.0, 0.1, 10.0])
    out["g9_db"] = dbs
    out["g9_level"] = lv
    out["g9_mult"] = np.array([syn.db_to_multiplier(d, x) for d, x in zip(dbs, lv)])

    # G5 normalize_irs
    irs5 = make_irs(rng, 4, 3, 500)
    out["g5_irs"] = irs5
    out["g5_norm"] = syn.normalize_irs(irs5.transpose(1, 0, 2)).transpose(1, 0, 2)

    # G1 static event
    a1 = make_clip(rng, 8000)
    h1 = make_irs(rng, 4, 1, 2000)
    ev = FakeEvent("g1", a1, 1, snr=10.0, sr=sr)
    syn.render_event_audio(ev, h1, "mic000", ref_db=-65)
    out.update(g1_audio=a1, g1_irs=h1, g1_spatial=ev.spatial_audio["mic000"], g1_snr=10.0)
    out["g1_full_conv"] = syn.time_invariant_convolution(a1.astype(np.float64), h1[:, 0].T)

    # G1b static event longer IR than clip, negative-ish snr edge (snr small)
    a1b = make_clip(rng, 1500)
    h1b = make_irs(rng, 3, 1, 4000)
    ev = FakeEvent("g1b", a1b, 1, snr=0.5, sr=sr)
    syn.render_event_audio(ev, h1b, "mic000", ref_db=-50)
    out.update(g1b_audio=a1b, g1b_irs=h1b, g1b_spatial=ev.spatial_audio["mic000"])

    # G2 zero-emitter event (tile)
    a2 = make_clip(rng, 3000)
    ev = FakeEvent("g2", a2, 0, snr=7.0, sr=sr)
    syn.render_event_audio(ev, np.zeros((4, 0, 100)), "mic000", ref_db=-65)
    out.update(g2_audio=a2, g2_spatial=ev.spatial_audio["mic000"])

    # G3 moving events: N=3 and N=5 (N=5 has relevant.mean() < 0.5 masking active)
    for tag, n_ir, n_audio in (("g3a", 3, 6000), ("g3b", 5, 9000)):
        a3 = make_clip(rng, n_audio)
        h3 = make_irs(rng, 4, n_ir, 1200)
        ev = FakeEvent(tag, a3, n_ir, snr=12.0, sr=sr, is_moving=True)
        syn.render_event_audio(ev, h3, "mic000", ref_db=-65)
        w = syn.generate_interpolation_matrix(np.linspace(0, ev.duration, n_ir), sr, 128)
        hn = syn.normalize_irs(h3.transpose(1, 0, 2)).transpose(1, 0, 2)
        raw = syn.time_variant_convolution(hn, ev, 512, 256, 128)
        out.update({f"{tag}_audio": a3, f"{tag}_irs": h3, f"{tag}_spatial": ev.spatial_audio["mic000"],
                    f"{tag}_w": w, f"{tag}_raw": raw})

    # G4 dry path
    a4 = make_clip(rng, 4000)
    h4 = make_irs(rng, 4, 1, 1500)
    ev = FakeEvent("g4", a4, 1, snr=9.0, sr=sr, ref_ir_channel=0, direct_path_time_ms=[5, 60])
    syn.render_event_audio(ev, h4, "mic000", ref_db=-65)
    out.update(g4_audio=a4, g4_irs=h4, g4_spatial=ev.spatial_audio["mic000"],
               g4_dry=ev._spatial_audio_dry["mic000"])

    # G6 powerlaw noise
    for beta in (0, 1, 2, -1):
        for n in (1000, 1001):
            out[f"g6_b{beta}_n{n}"] = amb.powerlaw_psd_gaussian(beta, (4, n))
    out["g6_fmin"] = amb.powerlaw_psd_gaussian(1, (2, 512), fmin=0.1, seed=7)
    out["g6_1d"] = amb.powerlaw_psd_gaussian(1, 300)

    # G7 Ambience.load_ambience + the db multiplier
    am = amb.Ambience(channels=4, duration=0.5, alias="amb", noise="pink", ref_db=-60, sample_rate=sr)
    noise = am.load_ambience(normalize=True)
    out["g7_noise"] = noise
    out["g7_mult"] = np.array(syn.db_to_multiplier(am.ref_db, np.mean(np.abs(noise))))

    # G8 full mixdown: 5 events (one clipped at scene end, two overlapping, one moving, one dry) + white ambience
    dur = 2.0
    n_caps = 4
    specs = [  # (n_audio, n_emit, start, snr, moving, dry)
        (4000, 1, 0.10, 10.0, False, False),
        (4000, 1, 0.30, 20.0, False, False),   # overlaps the first
        (6000, 1, 1.60, 15.0, False, False),   # runs past the scene end -> clipped
        (5000, 3, 0.90, 8.0, True, False),     # moving
        (3000, 1, 1.00, 6.0, False, True),     # dry path
    ]
    events, ir_list = {}, []
    for i, (na, ne, st, snr, mv, dry) in enumerate(specs):
        a = make_clip(rng, na)
        h = make_irs(rng, n_caps, ne, 1000)
        ir_list.append(h)
        events[f"ev{i}"] = FakeEvent(f"ev{i}", a, ne, snr, sr, scene_start=st, is_moving=mv,
                                     ref_ir_channel=1 if dry else None,
                                     direct_path_time_ms=[2, 20] if dry else None)
        out[f"g8_audio{i}"] = a
        out[f"g8_irs{i}"] = h
    mic_ir = np.concatenate(ir_list, axis=1)
    ambience = amb.Ambience(channels=n_caps, duration=dur, alias="a0", noise="white", ref_db=-65, sample_rate=sr)
    state = types.SimpleNamespace(irs={"mic000": mic_ir}, get_irs=lambda: {"mic000": mic_ir},
                                  simulate=lambda: None, microphones={"mic000": object()},
                                  num_emitters=mic_ir.shape[1], name="fake")
    scene = types.SimpleNamespace(state=state, events=events, ambience={"a0": ambience}, audio={},
                                  ref_db=-65, duration=dur, sample_rate=sr)
    syn.render_audio_for_all_scene_events(scene)
    syn.generate_scene_audio_from_events(scene)
    out["g8_specs"] = np.array([[s[0], s[1], s[2], s[3], float(s[4]), float(s[5])] for s in specs])
    out["g8_scene"] = scene.audio["mic000"]
    out["g8_ambience"] = ambience.audio
    for i, ev in enumerate(events.values()):
        out[f"g8_spatial{i}"] = ev.spatial_audio["mic000"]
        out[f"g8_padded{i}"] = ev._spatial_audio_padded["mic000"].astype(np.float32)
    out["g8_dry4"] = events["ev4"]._spatial_audio_dry["mic000"]
    out["g8_dry_padded4"] = events["ev4"]._spatial_audio_dry_padded["mic000"]

    for k in list(out):
        if k.endswith("_irs") or "_irs" in k:
            assert np.array_equal(out[k].astype(np.float32).astype(np.float64), out[k])
            out[k] = out[k].astype(np.float32)
    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
    stft_vectors(syn)
    geometry_vectors(syn)
    edge_vectors(syn)


def stft_vectors(syn):
    """G10: the STFT-domain intermediates of the moving path (synthesize.py:109-274) on a small case of its own
    (separate seed and file, so reference_vectors.npz is untouched): stft of clip and IRs (both axis conventions),
    perform_time_variant_convolution, istft_overlap_synthesis."""
    rng = np.random.default_rng(20261003)
    sr, n_ir = 8000, 3
    a = make_clip(rng, 2000)
    h = make_irs(rng, 2, n_ir, 600)
    ev = FakeEvent("g10", a, n_ir, snr=10.0, sr=sr, is_moving=True)
    s_a = syn.stft(a.astype(np.float64))
    s_h = syn.stft(h)
    s_h_last = syn.stft(h, stft_dims_first=False)
    w = syn.generate_interpolation_matrix(np.linspace(0, ev.duration, n_ir), sr, 128)
    y = syn.perform_time_variant_convolution(s_a, s_h, w)
    x = syn.istft_overlap_synthesis(y, 512, 256, 128)
    # a second parameter set: fft 256 / win 128 / hop 64 on a 1-D signal
    s_b = syn.stft(a[:777].astype(np.float64), 256, 128, 64)
    x_b = syn.istft_overlap_synthesis(s_b[:, :, None], 256, 128, 64)
    out = dict(g10_audio=a, g10_irs=h.astype(np.float32), g10_w=w, g10_stft_audio=s_a.astype(np.complex64),
               g10_stft_irs=s_h.astype(np.complex64), g10_stft_irs_dims_last=s_h_last.astype(np.complex64),
               g10_tv=y.astype(np.complex64), g10_istft=x, g10_stft_b=s_b.astype(np.complex64), g10_istft_b=x_b)
    path = os.path.join(HERE, "reference_stft_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


def geometry_vectors(syn):
    """G14: moving events rendered with OTHER STFT geometries than the defaults (synthesize.py:507-516 exposes fft_size /
    win_size / hop_size and :277-310 honours any values the framing accepts): win != 2*hop, fft < 2*win - 1 (time-aliased
    frames), 75 % overlap, a scaled copy of the default geometry, non-power-of-two sizes, fft < win (frames cropped by
    rfft(n=fft_size)); plus the geometries the reference itself REFUSES
    (istft_overlap_synthesis needs fft <= 2*hop + win: numpy raises a broadcast ValueError otherwise; stft needs win >= hop:
    np.pad raises).  A file of its own with its own seed: the other golden files are untouched."""
    rng = np.random.default_rng(20261004)
    sr, n_ir, n_caps = 8000, 4, 3
    a = make_clip(rng, 6000)
    h = make_irs(rng, n_caps, n_ir, 1500)
    out = dict(g14_audio=a, g14_irs=h.astype(np.float32))
    geoms = [(512, 256, 192), (256, 256, 128), (384, 256, 64), (1024, 512, 256), (512, 384, 128), (300, 200, 100), (192, 256, 128)]
    out["g14_geometries"] = np.array(geoms)
    for fft_size, win, hop in geoms:
        ev = FakeEvent("g14", a, n_ir, snr=11.0, sr=sr, is_moving=True)
        syn.render_event_audio(ev, h, "mic000", ref_db=-65, fft_size=fft_size, win_size=win, hop_size=hop)
        hn = syn.normalize_irs(h.transpose(1, 0, 2)).transpose(1, 0, 2)
        raw = syn.time_variant_convolution(hn, ev, fft_size, win, hop)
        tag = f"g14_{fft_size}_{win}_{hop}"
        out[tag + "_spatial"] = ev.spatial_audio["mic000"]
        out[tag + "_raw"] = raw
    refused = []
    for fft_size, win, hop in [(1024, 512, 128), (512, 256, 64), (512, 128, 256)]:
        ev = FakeEvent("g14", a, n_ir, snr=11.0, sr=sr, is_moving=True)
        try:
            syn.render_event_audio(ev, h, "mic000", ref_db=-65, fft_size=fft_size, win_size=win, hop_size=hop)
        except ValueError:
            refused.append((fft_size, win, hop))
    out["g14_refused"] = np.array(refused)
    path = os.path.join(HERE, "reference_geometry_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays; refused by the reference:", refused)


def edge_vectors(syn):
    """G15: degenerate events as the REFERENCE renders them (its float64 arithmetic multiplies zeros by 1 / tiny and by
    10^(dB/20) / tiny and keeps zeros): snr = 0 (silence), a negative snr (polarity flips), an all-zero IR, an all-zero clip, a
    moving event with one all-zero IR among its emitters, a dry render of an all-zero clip.  A file and seed of its own."""
    rng = np.random.default_rng(20261005)
    sr, n_caps = 8000, 2
    a = make_clip(rng, 1500)
    h = make_irs(rng, n_caps, 1, 400)
    h3 = make_irs(rng, n_caps, 3, 400)
    h3[:, 1, :] = 0.0
    out = dict(g15_audio=a, g15_irs=h.astype(np.float32), g15_irs_moving=h3.astype(np.float32))
    cases = {"snr0": (a, h, 0.0, False, {}), "snr_neg": (a, h, -4.0, False, {}), "zero_ir": (a, np.zeros_like(h), 9.0, False, {}),
             "zero_clip": (np.zeros_like(a), h, 9.0, False, {}), "moving_one_zero_ir": (a, h3, 7.0, True, {}),
             "zero_clip_dry": (np.zeros_like(a), h, 9.0, False, dict(ref_ir_channel=0, direct_path_time_ms=[2, 20]))}
    for tag, (clip, irs, snr, moving, kw) in cases.items():
        ev = FakeEvent(tag, clip, irs.shape[1], snr=snr, sr=sr, is_moving=moving, **kw)
        with np.errstate(all="ignore"):
            syn.render_event_audio(ev, irs, "mic000", ref_db=-65)
        out[f"g15_{tag}_spatial"] = ev.spatial_audio["mic000"]
        if kw:
            out[f"g15_{tag}_dry"] = ev._spatial_audio_dry["mic000"]
    out["g15_snrs"] = np.array([0.0, -4.0, 9.0, 9.0, 7.0, 9.0])
    path = os.path.join(HERE, "reference_edge_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays;",
          {k: (bool(np.isfinite(v).all()), float(np.abs(v).max())) for k, v in out.items() if k.endswith(("_spatial", "_dry"))})


if __name__ == "__main__":
    main()
