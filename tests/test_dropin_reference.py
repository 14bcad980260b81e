"""The drop-in claim, checked against the REAL reference package where it is present (the build container):
``dropin.install()`` rebinds the functions ``audiblelight.core.Scene.generate`` resolves by lazy import
(reference core.py:1828-1831), and the reference module's own entry points then render the golden scene through
the (host-emulated) kernels.  Skipped on machines without /root/reference (the GPU box): nothing else reads it."""
import importlib.metadata
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import pytest

from audiblelight_amd import _hip, dropin, engine, synthesize as ours
from tests import hostemu
from tests.conftest import assert_parity

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "audiblelight")), reason="reference tree not present")


@pytest.fixture()
def reference_synthesize():
    """Import the reference's synthesize module with stand-ins for the third-party packages absent here
    (same recipe as tests/golden/make_golden.py); the stand-ins and the reference package are removed from sys.modules afterwards."""
    stubbed = []
    for name in ["librosa", "librosa.util", "librosa.effects", "soundfile", "trimesh", "trimesh.visual", "loguru", "deepdiff",
                 "pedalboard", "pysofaconventions", "rlr_audio_propagation", "rtree", "pyroomacoustics", "gdown", "h5py",
                 "cv2", "pyvista", "netCDF4", "vtk"]:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = MagicMock()
            stubbed.append(name)
    real_version = importlib.metadata.version
    importlib.metadata.version = lambda n: "0.1.2" if n == "audiblelight" else real_version(n)
    sys.path.insert(0, REF)
    try:
        import audiblelight.synthesize as ref_syn
        yield ref_syn
    finally:
        importlib.metadata.version = real_version
        sys.path.remove(REF)
        for name in stubbed + [m for m in sys.modules if m == "audiblelight" or m.startswith("audiblelight.")]:
            sys.modules.pop(name, None)


def test_install_rebinds_what_scene_generate_imports(reference_synthesize, golden):
    from tests.test_hostemu_api import build_g8_scene

    ref_syn = reference_synthesize
    originals = {n: getattr(ref_syn, n) for n in dropin.REPLACED}
    ours.set_renderer(engine.Renderer(lib=_hip.Library(hostemu.build()), memory=hostemu.NumpyMemory()))
    dropin.install(ref_syn)
    try:
        # the two names Scene.generate imports at call time now resolve to the MI355X path
        from audiblelight.synthesize import generate_scene_audio_from_events, render_audio_for_all_scene_events
        assert render_audio_for_all_scene_events is ours.render_audio_for_all_scene_events
        assert generate_scene_audio_from_events is ours.generate_scene_audio_from_events
        scene = build_g8_scene(golden)
        ref_syn.render_audio_for_all_scene_events(scene)
        ref_syn.generate_scene_audio_from_events(scene)
        assert scene.audio["mic000"].dtype == np.float32
        assert_parity(scene.audio["mic000"], golden["g8_scene"], 1e-4)
        # scalar helpers keep the reference's results (its own implementation is the check here)
        x = np.linspace(-0.5, 0.7, 101)
        np.testing.assert_allclose(ref_syn.apply_snr(x, 7.0), originals["apply_snr"](x, 7.0), rtol=1e-6)
        assert ref_syn.db_to_multiplier(-40.0, 0.3) == pytest.approx(originals["db_to_multiplier"](-40.0, 0.3), rel=1e-6)
    finally:
        dropin.uninstall(ref_syn)
        ours.set_renderer(None)
    for n in dropin.REPLACED:
        assert getattr(ref_syn, n) is originals[n]


def test_differential_fuzz_against_the_reference():
    """tests/golden/differential_fuzz.py on a handful of seeds: random scenes (static / moving / tiled events, snr 0 and negative,
    silent clips, all-zero IRs, events off the scene's ends, coloured ambiences, dry renders) rendered by the REFERENCE's own
    functions and by this package's over the host-emulated kernels, every array compared.  In a process of its own: the reference
    tree has a `tests` package too.  (600+ seeds: profiles/r05l_differential_fuzz.txt.)"""
    import subprocess

    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "differential_fuzz.py")
    res = subprocess.run([sys.executable, script, "0", "25"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "0 outside 1e-4" in res.stdout
