"""CPU oracle for the AudibleLight synthesis hot path.

TEST INFRASTRUCTURE ONLY.  This module is a float64 numpy/scipy restatement of the
reference algorithm.  It may be imported by ``tests/``, by ``__graft_entry__.smoke()`` and
by the ``cpu_baseline`` leg of ``bench.py`` -- never by the product package
``audiblelight_amd`` (which must fail loudly when the HIP extension is missing).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` (this container only) and stores its outputs on seeded inputs under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below against
those vectors (<=1e-10) and against the reference's own known-answer tests
(``tests/test_synthesize.py:42-57,307-337``).

Each function cites the reference lines it restates (paths relative to /root/reference).
"""

from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
from scipy import fft as sp_fft
from scipy import signal as sp_signal

FFT_SIZE, WIN_SIZE, HOP_SIZE = 512, 256, 128  # audiblelight/config.py:9-11
DEFAULT_REF_DB = -65  # audiblelight/config.py:23
SEED = 42  # audiblelight/utils.py:35


# --------------------------------------------------------------------------- helpers
def tiny(x) -> float:
    """Smallest normal of x's float dtype, float32 for non-float input (utils.py:691-706)."""
    dt = np.asarray(x).dtype
    if not (np.issubdtype(dt, np.floating) or np.issubdtype(dt, np.complexfloating)):
        dt = np.dtype(np.float32)
    return np.finfo(dt).tiny


def fit_length(x: np.ndarray, n: int, mode: str = "constant") -> np.ndarray:
    """(C, L) -> (C, n): truncate, or pad on the right (utils.py:667-688)."""
    have = x.shape[1]
    if have < n:
        return np.pad(x, ((0, 0), (0, n - have)), mode=mode)
    return x[:, :n] if have > n else x


def check_audio(x: np.ndarray) -> None:
    """What librosa.util.valid_audio enforces on the path (synthesize.py:398,552,603)."""
    if not isinstance(x, np.ndarray) or not np.issubdtype(x.dtype, np.floating):
        raise ValueError("Audio data must be floating-point ndarray")
    if x.ndim == 0 or not np.isfinite(x).all():
        raise ValueError("Audio buffer is not finite everywhere")


# --------------------------------------------------------------------------- A4 / A5 / A9
def snr_scale(x: np.ndarray, snr: float) -> np.ndarray:
    """x * snr / max(|x|, 1e-15)  (synthesize.py:40-49)."""
    peak = max(float(np.abs(x).max()) if x.size else 0.0, 1e-15)
    return x * snr / peak


def db_gain(db: float, level) -> float:
    """10^(db/20) / (level + tiny(level))  (synthesize.py:52-68)."""
    return 10.0 ** (db / 20.0) / (level + tiny(level))


def level_law(x: np.ndarray, snr: float, ref_db: float):
    """apply_snr then db_to_multiplier, as chained at synthesize.py:594-599.

    Returns (scaled_audio, event_scale) where event_scale is the second multiplier only
    (the one compute_dry_audio receives, synthesize.py:598,608).
    """
    y = snr_scale(x, snr)
    event_scale = db_gain(ref_db + snr, np.mean(np.abs(y)))
    return event_scale * y, event_scale


# --------------------------------------------------------------------------- A1
def unit_energy_irs(irs: np.ndarray) -> np.ndarray:
    """Divide by the mean (over axis -2) L2 norm (over axis -1)  (synthesize.py:404-428)."""
    norm = np.sqrt((np.abs(irs) ** 2).sum(axis=-1, keepdims=True))
    norm = norm + tiny(norm)
    return irs / norm.mean(axis=-2, keepdims=True)


def emitter_gains(irs_cnl: np.ndarray) -> np.ndarray:
    """Per-emitter scalar that render_event_audio's normalisation applies (synthesize.py:560).

    irs_cnl is (C, N, L); the reference transposes to (N, C, L) so the mean runs over
    capsules.  Returns g[N] with irs_norm[:, n] = irs[:, n] * g[n].
    """
    e = np.sqrt((irs_cnl.astype(np.float64) ** 2).sum(axis=-1))  # (C, N)
    e = e + tiny(e)
    return 1.0 / e.mean(axis=0)


# --------------------------------------------------------------------------- A2
def convolve_static(audio: np.ndarray, ir_lc: np.ndarray) -> np.ndarray:
    """Full linear convolution of a mono clip with each capsule IR (synthesize.py:71-106)."""
    if audio.ndim != 1:
        raise ValueError(f"Only mono input is supported, but got {audio.ndim} dimensions!")
    if ir_lc.ndim != 2:
        raise ValueError(
            f"Expected shape of IR should be (n_samples, n_channels), but got ({ir_lc.shape}) instead"
        )
    return sp_signal.fftconvolve(audio[:, None], ir_lc, mode="full", axes=0).T


# --------------------------------------------------------------------------- A7
def sin2_window(win: int = WIN_SIZE) -> np.ndarray:
    return np.sin(np.pi / win * np.arange(win)) ** 2  # synthesize.py:120


def frame_count(n: int, hop: int = HOP_SIZE) -> int:
    return 2 * int(np.ceil(n / (2.0 * hop))) + 1  # synthesize.py:123


def stft_frames(y: np.ndarray, nfft=FFT_SIZE, win=WIN_SIZE, hop=HOP_SIZE) -> np.ndarray:
    """sin^2-windowed rFFT frames, frame/frequency axes first (synthesize.py:109-145)."""
    nfr = frame_count(y.shape[-1], hop)
    lead = [(0, 0)] * (y.ndim - 1)
    yp = np.pad(y, lead + [(win - hop, nfr * hop - y.shape[-1])])
    idx = np.arange(win)[:, None] + hop * np.arange(nfr)[None, :]  # (win, frames)
    frames = yp[..., idx] * sin2_window(win)[:, None]
    spec = sp_fft.rfft(frames, nfft, axis=-2)  # (..., freq, frames)
    return np.ascontiguousarray(np.moveaxis(np.moveaxis(spec, -2, 0), -1, 0))


def crossfade_weights(ir_times: np.ndarray, sr: float, hop: int = HOP_SIZE,
                      n_frames: Optional[int] = None) -> np.ndarray:
    """Linear cross-fade weights W[frame, ir]  (synthesize.py:148-181)."""
    starts = np.round((np.asarray(ir_times, dtype=float) * sr + hop) / hop)
    rows = int(starts[-1]) if n_frames is None else n_frames
    w = np.zeros((rows, len(starts)))
    for a in range(len(starts) - 1):
        span = np.arange(starts[a], starts[a + 1] + 1, dtype=int) - 1
        ramp = np.linspace(0.0, 1.0, len(span))
        w[span, a] = 1.0 - ramp
        w[span, a + 1] = ramp
    return w


def tv_frames(n_audio: int, duration: float, n_irs: int, sr: float, hop: int = HOP_SIZE):
    """(W, n_frames) used by time_variant_convolution (synthesize.py:302-303,208-210)."""
    w = crossfade_weights(np.linspace(0, duration, n_irs), sr, hop)
    return w, min(frame_count(n_audio, hop), w.shape[0])


def convolve_moving_stft(audio: np.ndarray, irs_cnl: np.ndarray, duration: float, sr: float,
                         nfft=FFT_SIZE, win=WIN_SIZE, hop=HOP_SIZE) -> np.ndarray:
    """STFT-domain time-variant convolution, restated literally (synthesize.py:184-310).

    Y[i] = sum_{k<=min(i,F_ir-1)} S[i-k] * sum_l W[i-k,l] H[k,:,:,l];  un-normalised irFFT
    (norm="forward" => x fft size), overlap-add at hop, slice [win : n_frames*hop].
    O(F_a * F_ir) spectra products: small cases only.
    """
    h = stft_frames(irs_cnl, nfft, win, hop)  # (F_ir, freq, C, N)
    s = stft_frames(audio, nfft, win, hop)  # (F_a, freq)
    w, n_frames = tv_frames(audio.shape[-1], duration, irs_cnl.shape[1], sr, hop)
    f_ir, n_freq, n_ch, _ = h.shape
    out_spec = np.zeros((n_frames, n_freq, n_ch), dtype=complex)
    for i in range(n_frames):
        for k in range(min(i, f_ir - 1) + 1):
            mixed = h[k] @ w[i - k].astype(complex)  # (freq, C)
            out_spec[i] += mixed * s[i - k][:, None]
    frames = np.real(sp_fft.irfft(out_spec, n=nfft, axis=1, norm="forward"))
    ola = np.zeros(((n_frames + 1) * hop + win, n_ch))
    for i in range(n_frames):
        ola[i * hop: i * hop + nfft] += frames[i]
    return ola[win: n_frames * hop].T


def crossfade_envelopes(w: np.ndarray, n_frames: int, n_audio: int,
                        win=WIN_SIZE, hop=HOP_SIZE) -> np.ndarray:
    """env[l, t] = sum_{j<n_frames} W[j,l] * win(t + (win-hop) - hop*j), t in [0, n_audio)."""
    window = sin2_window(win)
    env = np.zeros((w.shape[1], n_audio + win))
    for j in range(n_frames):
        lo = hop * j - (win - hop)
        a, b = max(lo, 0), min(lo + win, n_audio)
        if b > a:
            env[:, a:b] += w[j][:, None] * window[a - lo: b - lo][None, :]
    return env[:, :n_audio]


def convolve_moving(audio: np.ndarray, irs_cnl: np.ndarray, duration: float, sr: float,
                    nfft=FFT_SIZE, win=WIN_SIZE, hop=HOP_SIZE) -> np.ndarray:
    """Envelope form of time_variant_convolution.

    Because the sin^2 window at 50 % overlap is a partition of unity and a 512-point FFT
    holds the linear convolution of two 256-sample frames, the STFT-domain algorithm equals
        y_c = nfft * sum_l fftconvolve(audio * env_l, h_{c,l})[: n_frames*hop - win].
    ``tests/test_oracle_golden.py`` checks this against ``convolve_moving_stft`` and the
    reference output.  This is the form the HIP path computes.
    """
    w, n_frames = tv_frames(audio.shape[-1], duration, irs_cnl.shape[1], sr, hop)
    env = crossfade_envelopes(w, n_frames, audio.shape[-1], win, hop)
    n_out = max(n_frames * hop - win, 0)
    out = np.zeros((irs_cnl.shape[0], n_out))
    for l in range(irs_cnl.shape[1]):
        src = audio.astype(np.float64) * env[l]
        if not src.any():
            continue
        full = sp_signal.fftconvolve(src[None, :], irs_cnl[:, l, :], mode="full", axes=1)
        out += nfft * full[:, :n_out] if full.shape[1] >= n_out else \
            nfft * np.pad(full, ((0, 0), (0, n_out - full.shape[1])))
    return out


# --------------------------------------------------------------------------- A8
def dry_path(audio: np.ndarray, irs_norm_cnl: np.ndarray, event_scale: float,
             ref_channel: int, window_ms: Sequence[float], sr: float) -> np.ndarray:
    """Direct-path render on one reference capsule (synthesize.py:432-504)."""
    if ref_channel > irs_norm_cnl.shape[0]:
        raise ValueError(
            f"Reference channel index out of range for IRs with {irs_norm_cnl.shape[0]} channels")
    lo = int(window_ms[0] * sr / 1000)
    hi = int(window_ms[1] * sr / 1000)
    ir = irs_norm_cnl[ref_channel, 0, :].copy()
    peak = int(np.argmax(ir))
    if peak + hi < ir.shape[0]:
        ir[peak + hi:] = 0
    if peak - lo > 0:
        ir[: peak - lo] = 0
    return sp_signal.fftconvolve(audio, ir, mode="full", axes=0) * event_scale


# --------------------------------------------------------------------------- A6
def render_event(audio: np.ndarray, irs_cnl: np.ndarray, snr: float, ref_db: float = DEFAULT_REF_DB,
                 is_moving: bool = False, duration: Optional[float] = None, sr: float = 44100,
                 ref_ir_channel: Optional[int] = None, direct_path_time_ms=None,
                 moving_impl: str = "envelope", nfft=FFT_SIZE, win=WIN_SIZE, hop=HOP_SIZE) -> dict:
    """One event at one microphone (synthesize.py:507-608). audio is the loaded, peak-normalised clip.
    nfft / win / hop: the STFT geometry of a moving event (synthesize.py:514-516); the envelope form only holds for
    win == 2*hop and nfft >= 2*win - 1, any other geometry goes through the literal STFT-domain restatement."""
    check_audio(audio)
    n_ch, n_emit, _ = irs_cnl.shape
    n_audio = audio.shape[0]
    irs_n = unit_energy_irs(irs_cnl.copy().transpose(1, 0, 2)).transpose(1, 0, 2)
    if n_emit == 1:
        if is_moving:
            raise ValueError("Moving Event has only one emitter!")
        wet = convolve_static(audio, irs_n[:, 0].T)
    elif n_emit == 0:
        wet = np.repeat(audio[:, None], n_ch, 1).T
    else:
        if not is_moving:
            raise ValueError("Expected a moving event!")
        envelope_ok = win == 2 * hop and nfft >= 2 * win - 1
        fn = convolve_moving if (moving_impl == "envelope" and envelope_ok) else convolve_moving_stft
        wet = fn(audio, irs_n, duration, sr, nfft, win, hop)
    wet = fit_length(wet, n_audio)
    out, event_scale = level_law(wet, snr, ref_db)
    check_audio(out)
    res = dict(spatial=out, event_scale=event_scale, dry=None)
    if ref_ir_channel is not None and direct_path_time_ms is not None:
        res["dry"] = dry_path(audio, irs_n, event_scale, ref_ir_channel, direct_path_time_ms, sr)
    return res


# --------------------------------------------------------------------------- A11
def event_slot(scene_start: float, scene_end: float, sr: float, n_scene: int):
    """(start, end) sample slot; Python round() = banker's rounding (synthesize.py:361-362)."""
    return max(0, round(scene_start * sr)), min(round(scene_end * sr), n_scene)


def mix_scene(spatials: Sequence[np.ndarray], slots: Sequence[tuple], duration: float, sr: float,
              ambiences: Sequence[tuple] = (), dries: Optional[Sequence] = None,
              keep_padded: bool = True) -> dict:
    """Additive mixdown into a float32 (C, round(T*sr)) buffer (synthesize.py:314-401).

    ``slots`` are (scene_start_s, scene_end_s); ``ambiences`` are (normalised_audio, ref_db).
    """
    n_ch = max(s.shape[0] for s in spatials)
    n_scene = round(duration * sr)
    scene = np.zeros((n_ch, n_scene), dtype=np.float32)
    for noise, amb_db in ambiences:
        if noise.shape != scene.shape:
            raise ValueError(
                f"Scene ambient noise does not match expected shape. "
                f"Expected {scene.shape}, but got {noise.shape}.")
        scene += db_gain(amb_db, np.mean(np.abs(noise))) * noise
    padded, dry_padded = [], []
    for i, (x, (t0, t1)) in enumerate(zip(spatials, slots)):
        a, b = event_slot(t0, t1, sr, n_scene)
        if b <= a:
            padded.append(None)
            dry_padded.append(None)
            continue
        piece = fit_length(x, b - a)
        scene[:, a:b] += piece
        if keep_padded:
            full = np.zeros_like(scene)
            full[:, a:b] += piece
            padded.append(full)
        else:
            padded.append(None)
        if dries is not None and dries[i] is not None:
            d = np.zeros(n_scene, dtype=scene.dtype)
            d[a:b] += fit_length(dries[i][None, :], b - a)[0]
            dry_padded.append(d)
        else:
            dry_padded.append(None)
    check_audio(scene)
    return dict(scene=scene, padded=padded, dry_padded=dry_padded)


# --------------------------------------------------------------------------- A12
def powerlaw_draws(shape, seed: int = SEED, scale=1.0):
    """The two normal draws powerlaw_psd_gaussian makes, in order (ambience.py:351-356)."""
    rng = np.random.default_rng(seed)
    size = list(shape)
    size[-1] = size[-1] // 2 + 1
    re = rng.normal(scale=scale, size=size)
    im = rng.normal(scale=scale, size=size)
    return re, im


def powerlaw_scale(beta: float, samples: int, fmin: float = 0.0):
    """Spectral shaping vector and output std (ambience.py:319-340)."""
    f = np.fft.rfftfreq(samples)
    if not 0 <= fmin <= 0.5:
        raise ValueError(f"Argument `fmin` must be chosen between 0 and 0.5 but got {fmin:.2f}.")
    fmin = max(fmin, 1.0 / (samples + tiny(samples)))
    s = f.copy()
    cut = int(np.sum(s < fmin))
    if cut and cut < len(s):
        s[:cut] = s[cut]
    s = s ** (-beta / 2.0)
    w = s[1:].copy()
    w[-1] *= (1 + (samples % 2)) / 2.0
    sigma = 2 * np.sqrt(np.sum(w ** 2)) / (samples + tiny(samples))
    return s, sigma


def powerlaw_noise(beta: float, shape, fmin: float = 0.0, seed: int = SEED) -> np.ndarray:
    """Timmer-Koenig (1/f)^beta gaussian noise (ambience.py:271-375)."""
    size = [shape] if isinstance(shape, (int, np.integer)) else list(shape)
    samples = size[-1]
    s, sigma = powerlaw_scale(beta, samples, fmin)
    re, im = powerlaw_draws(size, seed, scale=s)
    if samples % 2 == 0:
        im[..., -1] = 0
        re[..., -1] *= np.sqrt(2)
    im[..., 0] = 0
    re[..., 0] *= np.sqrt(2)
    return np.fft.irfft(re + 1j * im, n=samples, axis=-1) / sigma


def peak_normalise_rows(x: np.ndarray) -> np.ndarray:
    """Row-wise x / max(|x| + tiny)  (ambience.py:211-214)."""
    out = x.copy()
    for c in range(out.shape[0]):
        out[c] = out[c] / np.max(np.abs(out[c]) + tiny(out[c]))
    return out


def ambience_noise(beta, channels: int, duration: float, sr: int, normalize: bool = True,
                   gaussian_draws: Optional[np.ndarray] = None, **kw) -> np.ndarray:
    """Ambience.load_ambience for the synthetic-noise modes (ambience.py:142-217)."""
    n = round(duration * sr)
    if beta == "gaussian":
        out = gaussian_draws if gaussian_draws is not None else np.random.normal(0, 1, (channels, n))
        out = np.array(out, dtype=float)
    else:
        out = powerlaw_noise(beta, (channels, n), **kw)
    return peak_normalise_rows(out) if normalize else out


def tile_ambience(clip: np.ndarray, channels: int, n: int) -> np.ndarray:
    """File-mode tiling of a decoded clip to (channels, n) (ambience.py:176-208), mono/matching only."""
    clip = np.atleast_2d(clip)
    reps_c = 1
    if clip.shape[0] != channels:
        clip = clip[:1]
        reps_c = channels
    reps_t = -(-n // clip.shape[1])
    return np.tile(clip, (reps_c, reps_t))[:, :n]


# --------------------------------------------------------------------------- A13 / A14
def peak_normalise_clip(a: np.ndarray) -> np.ndarray:
    """a / max(|a| + tiny(a))  (event.py:535-536)."""
    return a / np.max(np.abs(a) + tiny(a))


def fx_wrap(fn, clip: np.ndarray) -> np.ndarray:
    """Augmentation.process contract: copy, apply, wrap-pad/truncate to input length (augmentation.py:91-130)."""
    out = fn(clip.copy())
    return fit_length(np.atleast_2d(out), max(clip.shape), mode="wrap")[0]


def fx_gain(a, gain_db):  # pedalboard.Gain: linear gain 10^(dB/20) (augmentation.py:1105-1136)
    return a * np.float32(10.0 ** (gain_db / 20.0))


def fx_invert(a):  # augmentation.py:1577-1580
    return np.negative(a)


def fx_reverse(a):  # augmentation.py:1598-1601
    return np.flip(a, axis=-1)


def _fade_curve(shape: str, ramp: np.ndarray, fade_in: bool) -> np.ndarray:
    pi = math.pi
    if fade_in:  # augmentation.py:1490-1508
        return {"linear": ramp,
                "exponential": np.power(2, ramp - 1) * ramp,
                "logarithmic": np.log10(0.1 + ramp) + 1,
                "quarter_sine": np.sin(ramp * pi / 2),
                "half_sine": np.sin(ramp * pi - pi / 2) / 2 + 0.5}[shape]
    return {"linear": 1 - ramp,  # augmentation.py:1510-1528
            "exponential": np.power(2, -ramp) * (1 - ramp),
            "logarithmic": np.log10(1.1 - ramp) + 1,
            "quarter_sine": np.sin(ramp * pi / 2 + pi / 2),
            "half_sine": np.sin(ramp * pi + pi / 2) / 2 + 0.5}[shape]


def fade_envelope(n: int, sr: int, in_len: float, out_len: float, in_shape: str, out_shape: str) -> np.ndarray:
    """Combined fade-in x fade-out gain curve (augmentation.py:1490-1554)."""
    n_in = min(int(round(in_len * sr)), n)
    n_out = min(int(round(out_len * sr)), n)
    g_in = np.ones(n)
    if n_in and in_shape != "none":
        g_in = np.clip(np.concatenate((_fade_curve(in_shape, np.linspace(0, 1, n_in), True),
                                       np.ones(n - n_in))), 0, 1)
    g_out = np.ones(n)
    if n_out and out_shape != "none":
        g_out = np.clip(np.concatenate((np.ones(n - n_out),
                                        _fade_curve(out_shape, np.linspace(0, 1, n_out), False))), 0, 1)
    return g_in * g_out


def fx_fade(a, sr, in_len, out_len, in_shape, out_shape):
    return a * fade_envelope(a.shape[-1], sr, in_len, out_len, in_shape, out_shape)


def fx_clipping(a, threshold_db):  # pedalboard.Clipping: hard clip at +-10^(dB/20) (augmentation.py:832-868)
    thr = np.float32(10.0 ** (threshold_db / 20.0))
    return np.clip(a, -thr, thr)


def fx_distortion(a, drive_db):  # pedalboard.Distortion: tanh waveshaper after a drive gain (augmentation.py:927-960)
    return np.tanh(np.float32(10.0 ** (drive_db / 20.0)) * a)


def fx_bitcrush(a, bit_depth):  # pedalboard.Bitcrush: round to 2^bits levels (augmentation.py:266-300)
    q = np.float32(2.0 ** bit_depth)
    return np.rint(a * q) / q


def fx_preemphasis(a, coef):
    """librosa.effects.preemphasis (0.11): lfilter([1,-coef],[1]) with zi = 2*y[0]-y[1] (augmentation.py:1385)."""
    a = np.asarray(a, dtype=np.float64)
    out, _ = sp_signal.lfilter([1.0, -coef], [1.0], a, zi=np.atleast_1d(2 * a[0] - a[1]))
    return out


def fx_deemphasis(a, coef):
    """librosa.effects.deemphasis (0.11): inverse filter from zero state minus the extrapolation term."""
    a = np.asarray(a, dtype=np.float64)
    out, _ = sp_signal.lfilter([1.0], [1.0, -coef], a, zi=np.zeros(1))
    return out - ((2 - coef) * a[0] - a[1]) / (3 - coef) * coef ** np.arange(len(a))


def fx_timewarp(a, sr, fps, decisions, mode):
    """TimeWarp* (augmentation.py:1604-1790).  ``decisions`` = the random()<prob outcomes per iterated row.

    The reference frames with librosa.util.frame (shape (frame_len, n_frames)) and iterates it by rows.
    """
    fl = round(sr / fps)
    n = len(a)
    if fl > n:
        sliced = a[None, :]
    else:
        nf = 1 + (n - fl) // fl
        sliced = np.stack([a[j * fl: j * fl + fl] for j in range(nf)], axis=-1)  # (fl, nf)
    rows = []
    for row, hit in zip(sliced, decisions):
        if mode == "silence":
            rows.append(np.zeros(len(row)) if hit else row)
        elif mode == "duplicate":
            rows.extend([row, row] if hit else [row])
        elif mode == "remove":
            if not hit:
                rows.append(row)
        elif mode == "reverse":
            rows.append(row[::-1] if hit else row)
    return np.concatenate(rows) if rows else a


# --------------------------------------------------------------------------- device generator witness (no reference counterpart)
def philox4x32_10(counter, key):
    """Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), plain Python
    integers: the counter-based generator the device ambience draws from (audiblelight_amd/csrc/al_rng.h).  The reference
    itself draws with numpy's PCG64 (ambience.py:351-356), which has no parallel form; this restatement exists so the tests
    can hold the kernel's draws to an independent implementation (and both to the Random123 known-answer vectors)."""
    c0, c1, c2, c3 = (int(x) & 0xFFFFFFFF for x in counter)
    k0, k1 = (int(x) & 0xFFFFFFFF for x in key)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = (p1 >> 32) ^ c1 ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ c3 ^ k1, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3
