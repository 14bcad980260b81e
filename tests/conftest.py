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


def set_switch(monkeypatch, name, value):
    """Set (or, value None, remove) one AL_* variable for this test and make the package parse its switches again: they are
    read ONCE per process (audiblelight_amd/switches.py), not on the per-scene path."""
    from audiblelight_amd import switches

    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, str(value))
    switches.reload()


@pytest.fixture(autouse=True)
def _fresh_switches():
    """Every test starts from the environment as it is (the previous test's monkeypatch has been undone by then)."""
    from audiblelight_amd import switches

    switches.reload()
    yield


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


# ----------------------------------------------------------------------------- the contract's parity metric, with margins
# SURVEY.md 8(d) "Parity metric": per scene and per event rms(y - y_ref) / rms(y_ref) <= 1e-4 AND
# max|y - y_ref| <= 1e-4 * max|y_ref|, y_ref = the float64 oracle.  Every GPU test asserts BOTH through assert_parity;
# the worst observed pair per test module is written to gpurun_out/r06_parity_margins.txt at the end of the session
# (copied to profiles/ by hand: gpurun_out/ is scratch).
PARITY_TOL = 1e-4
_MARGINS = {}


def parity_errors(got, ref):
    """(relative RMS error, max-abs error / max|ref|) of `got` against `ref` (float64)."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    peak = float(np.max(np.abs(ref))) if ref.size else 0.0
    worst = float(np.max(np.abs(got - ref))) if ref.size else 0.0
    return rel_rms(got, ref), worst / (peak if peak > 0 else 1.0)


def assert_parity(got, ref, tol=PARITY_TOL, what=None):
    """Both halves of the contract's bound; `what` labels the failure (event / row indices)."""
    assert np.shape(got) == np.shape(ref), (np.shape(got), np.shape(ref), what)
    rms, mx = parity_errors(got, ref)
    where = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[0]
    seen = _MARGINS.setdefault(where, [0.0, 0.0, 0])
    seen[0], seen[1], seen[2] = max(seen[0], rms), max(seen[1], mx), seen[2] + 1
    assert rms <= tol, ("relative RMS", rms, tol, what)
    assert mx <= tol, ("max-abs / max|ref|", mx, tol, what)
    return rms, mx


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        worker = os.environ.get("PYTEST_XDIST_WORKER")
        name = "r06_parity_margins" + (f"_{worker}" if worker else "") + ".txt"
        with open(os.path.join(out_dir, name), "w") as fh:
            fh.write("# worst observed parity errors per test module (tests/conftest.py::assert_parity); contract: both <= 1e-4\n")
            fh.write("# module                              comparisons   worst rel. RMS   worst max|err| / max|ref|\n")
            for where, (rms, mx, n) in sorted(_MARGINS.items()):
                fh.write(f"{where:<38s} {n:11d}   {rms:14.3e}   {mx:14.3e}\n")
    except OSError:
        pass
