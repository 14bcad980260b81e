"""TEST-ONLY host emulation of the HIP kernels (see hip/hip_runtime.h in this directory)."""
import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "_build", "libal_hostemu.so")
CSRC = os.path.join(ROOT, "audiblelight_amd", "csrc")
SRCS = [os.path.join(CSRC, "al_kernels.hip"), os.path.join(CSRC, "al_transforms.hip"), os.path.join(CSRC, "al_plan.cpp")]


def build(sanitize: bool = False) -> str:
    """Compile the kernel sources for the host (once per source change); pytest-xdist workers take turns behind a file lock."""
    import fcntl

    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(LIB), ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(sanitize)


def _build_locked(sanitize: bool) -> str:
    deps = SRCS + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", "audiblelight_hip.h"), os.path.join(HERE, "hip", "hip_runtime.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) > os.path.getmtime(d) for d in deps):
        return LIB
    flags = ["-std=c++17", "-O2", "-g", "-fPIC", "-pthread"] + (["-DHOSTEMU_THREADS", "-fsanitize=address,undefined"] if sanitize else [])
    objs = [os.path.join(os.path.dirname(LIB), os.path.basename(src) + ".o") for src in SRCS]
    jobs = [subprocess.Popen(["g++"] + flags + ["-x", "c++", "-I", HERE, "-c", src, "-o", obj]) for src, obj in zip(SRCS, objs)]
    if any(job.wait() != 0 for job in jobs):          # the two translation units compile side by side
        raise RuntimeError("host emulation build failed")
    subprocess.check_call(["g++", "-shared"] + flags + objs + ["-o", LIB])
    return LIB


class NumpyMemory:
    """'Device memory' for the host-emulated kernels: plain numpy arrays."""

    def stream(self):
        return ct.c_void_p(0)

    def empty(self, n, dtype=np.float32):
        return np.full(max(int(n), 1), np.nan if np.dtype(dtype).kind == "f" else 0, dtype=dtype)

    def zeros(self, n, dtype=np.float32):
        return np.zeros(max(int(n), 1), dtype=dtype)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.dtype.fields is not None:
            arr = arr.view(np.uint8)
        return arr.copy()

    def ptr(self, buf):
        return buf.ctypes.data

    def download(self, buf):
        return np.array(buf)

    def synchronize(self):
        pass
