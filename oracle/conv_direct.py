"""ctypes loader of oracle/conv_direct.c (TEST INFRASTRUCTURE ONLY): the direct-form, FFT-free C restatement of the
static convolution (synthesize.py:71-106 + utils.py:667-688) and of the float32 mixdown add (synthesize.py:373-378).
Built by ``make -C oracle`` (``__graft_entry__.build()``); used by tests/test_oracle_golden.py and the -m gpu parity tests
as a witness that shares no algorithm with either the numpy oracle or the HIP path."""
import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle_c.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "conv_direct.c")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ct.CDLL(LIB)
        _lib.al_oracle_conv_direct.argtypes = [ct.c_void_p, ct.c_long, ct.c_void_p, ct.c_long, ct.c_long, ct.c_void_p, ct.c_long]
        _lib.al_oracle_conv_direct.restype = None
        _lib.al_oracle_mix_add.argtypes = [ct.c_void_p, ct.c_long, ct.c_long, ct.c_void_p, ct.c_long, ct.c_long, ct.c_long]
        _lib.al_oracle_mix_add.restype = None
    return _lib


def conv_direct(audio: np.ndarray, ir_cl: np.ndarray, n_out: int) -> np.ndarray:
    """(C, n_out) float64: first n_out samples of the full convolution of a mono float32 clip with each IR row."""
    a = np.ascontiguousarray(audio, dtype=np.float32)
    h = np.ascontiguousarray(ir_cl, dtype=np.float64)
    out = np.zeros((h.shape[0], n_out), dtype=np.float64)
    load().al_oracle_conv_direct(a.ctypes.data, len(a), h.ctypes.data, h.shape[0], h.shape[1], out.ctypes.data, n_out)
    return out


def mix_add(scene: np.ndarray, x: np.ndarray, start: int, count: int) -> None:
    """scene[:, start:start+count] += x[:, :count] with one float32 rounding per add (in place)."""
    assert scene.dtype == np.float32 and scene.flags.c_contiguous
    xx = np.ascontiguousarray(x, dtype=np.float64)
    load().al_oracle_mix_add(scene.ctypes.data, scene.shape[0], scene.shape[1], xx.ctypes.data, xx.shape[1], start, count)
