import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the real reference on seeded inputs (made by tests/golden/make_golden.py)."""
    with np.load(GOLDEN) as z:
        return {k: z[k] for k in z.files}


def rel_rms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt(np.mean(b ** 2))
    return float(np.sqrt(np.mean((a - b) ** 2)) / (den if den > 0 else 1.0))


def pcm16(x):
    """float32 -> int16 as python-soundfile writes PCM_16 (libsndfile f2s_clip_array, clipping on): lrintf(x * 32768),
    saturated to [-32768, 32767]."""
    scaled = np.asarray(x, dtype=np.float32) * np.float32(32768.0)
    return np.clip(np.rint(scaled), -32768, 32767).astype(np.int16)
