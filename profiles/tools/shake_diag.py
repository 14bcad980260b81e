"""Diagnostic beside tests/test_gpu_shake.py: for every kernel family, how the schedule-perturbed builds differ from the product library
(count of differing elements, largest difference) and whether each build agrees with ITSELF run to run."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from audiblelight_amd import _hip, engine, switches  # noqa: E402
from tests import shake, test_gpu_shake as t  # noqa: E402


class Env:
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


mp = Env()
# arguments: names of tests/shake.py variants beyond s1 / s3w (e.g. s2 s4 s5 s6: built on demand, profiles/r06_shake_sweep.txt)
more = [a for a in sys.argv[1:] if a in shake.EXTRA_VARIANTS]
libs = {"product": engine.Renderer()}
for name, path in shake.existing_or_built(["s1", "s3w"] + more).items():
    libs[name] = engine.Renderer(lib=_hip.Library(path))
for fam in ([a for a in sys.argv[1:] if a in t.FAMILIES] or list(t.FAMILIES)):
    ref = t.render_family(libs["product"], fam, mp)
    first = None
    for name, r in libs.items():
        a, b = t.render_family(r, fam, mp), t.render_family(r, fam, mp)
        self_same = t.same(a, b)
        diffs = {k: (int(np.sum(a[k] != ref[k])), float(np.max(np.abs(a[k].astype(np.float64) - ref[k]))), float(np.max(np.abs(ref[k]))))
                 for k in ref if not np.array_equal(a[k], ref[k], equal_nan=True)}
        if name != "product" and first is None:
            first = a
        among = "" if name == "product" else f"  vs {list(libs)[1]}: {'bit-identical' if t.same(a, first) else 'DIFFERS'}"
        print(f"{fam:28s} {name:8s} self-consistent {self_same}{among}  vs product: {diffs if diffs else 'bit-identical'}", flush=True)
    switches.reload()
