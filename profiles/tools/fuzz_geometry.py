"""Lease-side evidence beside test_moving_events_under_other_stft_geometries: random STFT geometries (any the reference accepts:
win >= hop, fft <= 2*hop + win; fft_size a product of 2, 3, 5, 7 on even seeds, ANY integer on odd ones: Bluestein) x random moving events through render_event_audio on the MI355X,
every row against the oracle's literal STFT-domain restatement (pinned to the reference on seven geometries, G14).
    python3 profiles/tools/fuzz_geometry.py FIRST LAST"""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from audiblelight_amd import core, synthesize as syn   # noqa: E402
from oracle import synth_oracle as orc                  # noqa: E402
from tests.conftest import parity_errors                # noqa: E402

SMOOTH = sorted({2 ** a * 3 ** b * 5 ** c * 7 ** d for a in range(11) for b in range(4) for c in range(3) for d in range(2)
                 if 32 <= 2 ** a * 3 ** b * 5 ** c * 7 ** d <= 2048})


def case(seed):
    rng = np.random.default_rng(90_000 + seed)
    hop = int(rng.integers(16, 300))
    win = int(rng.integers(hop, 4 * hop + 1))
    ok = [f for f in SMOOTH if f <= 2 * hop + win]
    fft = int(rng.choice(ok[-12:])) if rng.random() < 0.7 else int(rng.choice(ok))     # mostly near the upper limit, some far below win
    if seed % 2:      # any size at all (primes included): al_stft / al_istft_ola through Bluestein's chirp-z
        fft = int(np.random.default_rng(86_000 + seed).integers(max(32, (2 * hop + win) // 3), min(1024, 2 * hop + win) + 1))
    n_irs, C = int(rng.integers(2, 7)), int(rng.integers(1, 5))
    La, Lir, sr = int(rng.integers(2000, 12000)), int(rng.integers(300, 3000)), int(rng.choice([8000, 16000, 44100]))
    a = rng.standard_normal(La).astype(np.float32)
    a /= np.abs(a).max()
    h = (rng.standard_normal((C, n_irs, Lir)) * np.exp(-np.arange(Lir) / (Lir / 5.0))).astype(np.float32)
    snr = float(rng.uniform(5, 30))
    ev = core.Event("f", a, sr, snr=snr, n_emitters=n_irs, is_moving=True)
    syn.render_event_audio(ev, h, "mic", ref_db=-60, fft_size=fft, win_size=win, hop_size=hop)
    got = ev.spatial_audio["mic"]
    want = orc.render_event(a, h.astype(np.float64), snr, ref_db=-60, is_moving=True, duration=La / sr, sr=sr, nfft=fft, win=win,
                            hop=hop)["spatial"]
    rms, mx = parity_errors(got, want)
    envelope = win == 2 * hop and fft >= 2 * win - 1
    return (fft, win, hop, n_irs, C, La, Lir), envelope, rms, mx


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    t0, worst, bad, env = time.time(), (0.0, 0.0, None), [], 0
    for seed in range(first, last):
        try:
            shape, envelope, rms, mx = case(seed)
        except Exception as exc:  # noqa: BLE001 -- a fuzz driver reports everything
            bad.append((seed, f"{type(exc).__name__}: {str(exc)[:160]}"))
            continue
        env += envelope
        if max(rms, mx) > max(worst[0], worst[1]):
            worst = (rms, mx, (seed,) + shape)
        if rms > 1e-4 or mx > 1e-4:
            bad.append((seed, shape, rms, mx))
    print(f"seeds {first}..{last - 1}: {last - first} moving events under random STFT geometries, {len(bad)} failed, {env} of them in the envelope "
          f"form, {time.time() - t0:.0f} s")
    print(f"worst rel. RMS {worst[0]:.2e}, max-abs / max|ref| {worst[1]:.2e} at (seed, fft, win, hop, n_irs, C, La, Lir) = {worst[2]}")
    for b in bad[:20]:
        print("FAILED", b)
    sys.exit(1 if bad else 0)
