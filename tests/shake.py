"""Schedule-perturbed builds of the HIP library (TEST INFRASTRUCTURE, never loaded by the package).

``-DAL_SHAKE=<seed>`` (audiblelight_amd/csrc/al_common.h) makes every wave sleep a wave-, workgroup- and site-dependent number of cycles
around every workgroup barrier, after every LDS-DMA issue and before every hand-counted ``s_waitcnt``: the inline-asm paths that
the host-emulation build -- and with it ASan / UBSan and the differential fuzz -- cannot see.  tests/test_gpu_shake.py renders one
batch per kernel family through each variant and asserts bit-identical output against the product library.

Variants (built in-tree under tests/shake_build/ by ``__graft_entry__.build()`` so that they travel to the GPU box like the product
.so; about as long to compile as the product library, all of them side by side):
  s1      AL_SHAKE=1: pseudo-random skews
  s3w     AL_SHAKE=3 (wave 0 always last to move on) + AL_Q16_WAVES=1 + AL_SPLIT_WAVES=2 (other register budgets / occupancies)
  revert  AL_SHAKE=3 + AL_TEST_REVERT_Q16_BARRIER: the round-4 LDS race of al_quad16.h re-introduced -- the variant the test must FAIL on
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "audiblelight_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "shake_build")
VARIANTS = {
    "s1": ["-DAL_SHAKE=1"],
    "s3w": ["-DAL_SHAKE=3", "-DAL_Q16_WAVES=1", "-DAL_SPLIT_WAVES=2"],
    "revert": ["-DAL_SHAKE=3", "-DAL_TEST_REVERT_Q16_BARRIER=1"],
}
# further skews, built on demand only (profiles/tools/shake_diag.py s2 s4 s5 s6: a wider sweep than the test suite's two variants)
EXTRA_VARIANTS = {
    "s2": ["-DAL_SHAKE=2"],
    "s4": ["-DAL_SHAKE=4"],                      # wave 0 always the FIRST to move on
    "s5": ["-DAL_SHAKE=5", "-DAL_Q16_WAVES=1"],
    "s6": ["-DAL_SHAKE=6", "-DAL_SPLIT_WAVES=2"],
}


def library_path(name: str) -> str:
    return os.path.join(OUT, f"libaudiblelight_hip_{name}.so")


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def build(names=None) -> dict:
    """Compile the named variants (default: all) with hipcc for gfx950; returns {name: path}.  Needs no GPU."""
    names = list(VARIANTS) if names is None else list(names)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp"))]
    deps += [os.path.join(ROOT, "include", "audiblelight_hip.h"), os.path.abspath(__file__)]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT, exist_ok=True)
    jobs, todo = [], []
    plan_obj = os.path.join(OUT, "al_plan.o")
    for name in names:
        if not _stale(library_path(name), deps):
            continue
        objs = []
        for src, extra in (("al_kernels.hip", []), ("al_transforms.hip", ["-fno-slp-vectorize"])):
            obj = os.path.join(OUT, f"{os.path.splitext(src)[0]}_{name}.o")
            jobs.append(subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + extra + {**VARIANTS, **EXTRA_VARIANTS}[name]
                                         + [os.path.join(CSRC, src), "-o", obj]))
            objs.append(obj)
        todo.append((name, objs))
    if todo:
        jobs.append(subprocess.Popen(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-c", os.path.join(CSRC, "al_plan.cpp"), "-o", plan_obj]))
        if any(j.wait() != 0 for j in jobs):
            raise RuntimeError("hipcc failed on a schedule-perturbed variant")
        for name, objs in todo:
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [plan_obj, "-o", library_path(name)])
            for o in objs:
                os.remove(o)
    return {name: library_path(name) for name in names}


def existing_or_built(names) -> dict:
    """The variants as __graft_entry__.build() left them in the tree (they travel to the GPU box with it; file times may not), compiled
    only where one is missing."""
    missing = [n for n in names if not os.path.exists(library_path(n))]
    if missing:
        build(missing)
    return {n: library_path(n) for n in names}
