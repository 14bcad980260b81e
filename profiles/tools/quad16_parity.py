"""Round 4: the B = 16384 quad-tile transforms (csrc/al_quad16.h) on the GPU against the oracle, every row: static events with
interior and edge windows, ragged IR tails, moving events (the rolled general signal path)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
_sw.set_env("AL_QUAD16", "1")
from audiblelight_amd import engine  # noqa: E402
from tests import mac_regimes as mr  # noqa: E402

r = engine.Renderer()
mr.run_static_case(r, 14, 3120301, 6.5, 2.5, C=3, E=2, expect_split=True, expect_quad=True)
mr.run_static_case(r, 14, 3121202, 14.3, 11.6, C=2, E=1, expect_split=True, expect_quad=True)
mr.run_moving_case(r, 14, 2.3, n_irs=6, k_mult=5.2, expect_moving=612, C=2, E=1)
print("quad16 parity ok")
