"""The C ABI from a plain C host (tests/c_caller/render_static.c, gcc, no Python / torch / C++ in the process): static
events rendered with al_render_batch + al_mixdown, every row compared with the float64 oracle.  The same comparison as
tests/test_gpu_parity.py makes through ctypes -- here to pin that include/audiblelight_hip.h alone is enough to bind
the library (struct layouts, buffer sizes, call order)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import synth_oracle as orc
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
TOL = 1e-4
EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c_caller", "_build", "render_static")


@pytest.mark.parametrize("C,E,La,Lir,log2_block", [(3, 2, 9001, 2500, 10), (4, 3, 50_000, 30_001, 13)])
def test_c_host_renders_what_the_oracle_renders(tmp_path, C, E, La, Lir, log2_block):
    if not os.path.exists(EXE):
        import __graft_entry__

        __graft_entry__.build_c_caller()
    rng = np.random.default_rng(C * 1000 + E)
    sr, ref_db = 48000, -65.0
    snr = rng.uniform(5, 30, E).astype(np.float32)
    clips = rng.standard_normal((E, La)).astype(np.float32)
    clips /= np.abs(clips).max(axis=1, keepdims=True)
    irs = (rng.standard_normal((C, E, Lir)) * np.exp(-np.arange(Lir) / (Lir / 6.0))).astype(np.float32)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(src, "wb") as f:
        np.array([C, E, La, Lir, log2_block], dtype=np.int32).tofile(f)
        np.array([ref_db], dtype=np.float32).tofile(f)
        snr.tofile(f)
        clips.tofile(f)
        irs.tofile(f)
    run = subprocess.run([EXE, str(src), str(dst)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert f"rendered {E} events x {C} capsules" in run.stdout
    out = np.fromfile(dst, dtype=np.float32)
    scale, spatial, scene = out[:E], out[E: E + E * C * La].reshape(E, C, La), out[E + E * C * La:].reshape(C, La)
    want_scene = np.zeros((C, La))
    for e in range(E):
        want = orc.render_event(clips[e], irs[:, [e], :].astype(np.float64), float(snr[e]), ref_db=ref_db, sr=sr)["spatial"]
        got = spatial[e] * scale[e]
        for c in range(C):
            assert_parity(got[c], want[c], TOL, what=(e, c))
        want_scene += want
    assert_parity(scene, want_scene, TOL)
