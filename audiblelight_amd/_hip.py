"""ctypes binding of the C ABI declared in include/audiblelight_hip.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is
no fallback of any kind: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as ct
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libaudiblelight_hip.so")

AL_OK, AL_E_BADARG, AL_E_HIP, AL_E_UNSUPPORTED = 0, -1, -2, -3
ABI_VERSION = 6   # AL_ABI_VERSION of include/audiblelight_hip.h these struct mirrors were written against
MIN_LOG2_BLOCK, MAX_LOG2_BLOCK = 10, 14
FLAG_NO_IR_NORM = 1
FLAG_SPLIT_SPECTRA = 32
FLAG_STATIC_MAC = 64
FLAG_ONLY_STATIC = 128
FLAG_NARROW_FFT = 4
FLAG_QUAD_SPECTRA = 256
# bits of al_batch.flags that only pick between equivalent code paths (narrow FFT, runs of blocks per workgroup)
DEBUG_FLAG_MASK = FLAG_NARROW_FFT | (7 << 12) | (0xff << 16) | (0x7f << 24)   # bit 12: static accumulate, one k-tile per workgroup

# numpy mirrors of al_event / al_stream (the tables are built on the host and copied to HBM)
EVENT_DTYPE = np.dtype([
    ("audio_off", "<i8"), ("out_off", "<i8"), ("len", "<i4"), ("valid_len", "<i4"), ("n_blocks", "<i4"),
    ("stream0", "<i4"), ("n_streams", "<i4"), ("yspec_base", "<i4"), ("part_base", "<i4"),
    ("snr", "<f4"), ("ref_db", "<f4"), ("reserved", "<i4")], align=True)
STREAM_DTYPE = np.dtype([
    ("event", "<i4"), ("emitter", "<i4"), ("j_lo", "<i4"), ("n_j", "<i4"), ("xspec_base", "<i4"),
    ("w_off", "<i4"), ("w_len", "<i4"), ("gain", "<f4")], align=True)
assert EVENT_DTYPE.itemsize == 56 and STREAM_DTYPE.itemsize == 32


class _Versioned(ct.Structure):
    """Descriptor with ``struct_size`` / ``abi_version`` at its head, filled in here so that the library can refuse a
    descriptor laid out for another header version instead of misreading it."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_size, self.abi_version = ct.sizeof(type(self)), ABI_VERSION


class AlBatch(_Versioned):
    _fields_ = [
        ("struct_size", ct.c_int32), ("abi_version", ct.c_int32),
        ("log2_block", ct.c_int32), ("n_capsules", ct.c_int32), ("n_events", ct.c_int32), ("n_streams", ct.c_int32),
        ("n_emitters", ct.c_int32), ("ir_len", ct.c_int32), ("ir_stride_c", ct.c_int64), ("ir_stride_n", ct.c_int64),
        ("n_partitions", ct.c_int32), ("max_blocks", ct.c_int32), ("max_nj", ct.c_int32), ("hop", ct.c_int32),
        ("event0", ct.c_int32), ("stream0", ct.c_int32), ("emitter0", ct.c_int32), ("xspec_block0", ct.c_int32),
        ("yspec_block0", ct.c_int32), ("flags", ct.c_int32),
        ("twiddle", ct.c_void_p), ("audio", ct.c_void_p), ("ir", ct.c_void_p), ("wtab", ct.c_void_p),
        ("events", ct.c_void_p), ("streams", ct.c_void_p),
        ("ir_energy", ct.c_void_p), ("emitter_gain", ct.c_void_p), ("hspec", ct.c_void_p), ("xspec", ct.c_void_p),
        ("yspec", ct.c_void_p), ("spatial", ct.c_void_p), ("partials", ct.c_void_p), ("event_stats", ct.c_void_p),
        ("event_scale", ct.c_void_p), ("clip_scale", ct.c_void_p),
        ("xspec_zero_block", ct.c_int32), ("hspec_zero_block", ct.c_int32), ("emitter_parts", ct.c_void_p),
    ]


class AlMix(_Versioned):
    _fields_ = [
        ("struct_size", ct.c_int32), ("abi_version", ct.c_int32),
        ("n_capsules", ct.c_int32), ("n_samples", ct.c_int32), ("tile", ct.c_int32), ("n_tiles", ct.c_int32),
        ("accumulate", ct.c_int32), ("reserved", ct.c_int32),
        ("tile_ptr", ct.c_void_p), ("tile_events", ct.c_void_p), ("slot_src", ct.c_void_p), ("slot_len", ct.c_void_p),
        ("slot_start", ct.c_void_p), ("slot_count", ct.c_void_p), ("slot_rows", ct.c_void_p), ("slot_event", ct.c_void_p),
        ("spatial", ct.c_void_p), ("event_scale", ct.c_void_p), ("scene", ct.c_void_p),
        ("ambience", ct.c_void_p), ("ambience_scale", ct.c_void_p),
    ]


class AlEventSpec(ct.Structure):      # al_event_spec
    _fields_ = [("n_samples", ct.c_int32), ("n_emitters", ct.c_int32), ("emitter0", ct.c_int32), ("is_moving", ct.c_int32),
                ("snr", ct.c_float), ("ref_db", ct.c_float), ("gain", ct.c_float), ("stft_len", ct.c_int32), ("duration", ct.c_double)]


class AlPlanInfo(ct.Structure):       # al_plan_info
    _fields_ = [(n, ct.c_int32) for n in ("log2_block", "n_capsules", "ir_len", "n_events", "n_streams", "n_emitters", "n_partitions",
                                          "hop", "fft_size", "max_blocks", "max_nj", "max_nj_sliding", "xspec_blocks", "yspec_blocks",
                                          "n_partials", "reserved")] + \
               [(n, ct.c_int64) for n in ("hspec_blocks", "audio_floats", "spatial_floats", "wtab_floats")]


class AlMixTables(ct.Structure):      # al_mix_tables
    _fields_ = [(n, ct.c_int32) for n in ("n_capsules", "n_samples", "tile", "n_tiles", "n_slots", "n_tile_events", "n_skipped", "reserved")] + \
               [(n, ct.c_void_p) for n in ("tile_ptr", "tile_events", "slot_src", "slot_len", "slot_start", "slot_count", "slot_rows",
                                           "slot_event", "skipped")]


class AlChunk(ct.Structure):          # al_chunk
    _fields_ = [(n, ct.c_int32) for n in ("event0", "n_events", "stream0", "n_streams", "emitter0", "n_emitters", "xspec_block0",
                                          "xspec_blocks", "yspec_block0", "yspec_blocks", "max_blocks", "max_nj")]


class HipError(RuntimeError):
    """A C-ABI call returned a negative status."""


# every symbol include/audiblelight_hip.h declares: (restype, argtypes)
_P, _S = ct.c_void_p, ct.c_void_p
SYMBOLS = {
    "al_last_error": (ct.c_char_p, []),
    "al_abi_version": (ct.c_int, []),
    "al_twiddle_bytes": (ct.c_int64, [ct.c_int]),
    "al_twiddle_init": (ct.c_int, [_P, ct.c_int, _S]),
    "al_ir_spectra": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_emitter_gains": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_emitter_norm_sums": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_emitter_gains_from_sums": (ct.c_int, [ct.POINTER(AlBatch), ct.c_int32, _S]),
    "al_forward_spectra": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_signal_spectra": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_spectral_mac": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_spectral_mac_variant": (ct.c_int, [ct.POINTER(AlBatch), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]),
    "al_plan_last_error": (ct.c_char_p, []),
    "al_plan_abi_version": (ct.c_int, []),
    "al_choose_log2_block": (ct.c_int32, [ct.c_int32, ct.c_int32]),
    "al_stft_frame_count": (ct.c_int32, [ct.c_int64, ct.c_int32]),
    "al_interpolation_rows": (ct.c_int32, [ct.c_int32, ct.c_double, ct.c_double, ct.c_int32]),
    "al_interpolation_matrix": (ct.c_int, [ct.c_int32, ct.c_double, ct.c_double, ct.c_int32, ct.c_int32, _P]),
    "al_plan_create": (ct.c_int, [_P, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_double, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_int32,
                                  ct.POINTER(ct.c_void_p)]),
    "al_plan_destroy": (None, [_P]),
    "al_plan_get_info": (ct.c_int, [_P, ct.POINTER(AlPlanInfo)]),
    "al_plan_events": (ct.c_void_p, [_P]),
    "al_plan_streams": (ct.c_void_p, [_P]),
    "al_plan_wtab": (ct.c_void_p, [_P]),
    "al_plan_audio_offsets": (ct.c_void_p, [_P]),
    "al_workspace_bytes": (ct.c_int64, [_P]),
    "al_plan_chunk": (ct.c_int, [_P, ct.c_int32, ct.c_int32, ct.POINTER(AlChunk)]),
    "al_plan_emitter_parts": (ct.c_int, [_P, _P]),
    "al_plan_batch_flags": (ct.c_int, [_P, ct.POINTER(AlChunk), ct.POINTER(ct.c_int32)]),
    "al_plan_mixdown": (ct.c_int, [_P, _P, _P, _P, _P, _P, ct.c_int32, ct.c_double, ct.c_double, ct.c_int32, ct.c_int32,
                                   ct.POINTER(ct.c_void_p)]),
    "al_mix_plan_destroy": (None, [_P]),
    "al_mix_plan_get": (ct.c_int, [_P, ct.POINTER(AlMixTables)]),
    "al_block_synthesis": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_event_levels": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_event_stats": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_event_levels_from_stats": (ct.c_int, [ct.POINTER(AlBatch), ct.c_int32, _S]),
    "al_render_batch": (ct.c_int, [ct.POINTER(AlBatch), _S]),
    "al_mixdown": (ct.c_int, [ct.POINTER(AlMix), _S]),
    "al_scale_rows": (ct.c_int, [_P, ct.c_int64, _P, _S]),
    "al_scale_rows_f64": (ct.c_int, [_P, ct.c_int64, _P, _S]),
    "al_clip_scales": (ct.c_int, [ct.POINTER(AlBatch), _P, _P, _S]),
    "al_peak_scale": (ct.c_int, [_P, ct.c_int64, ct.c_float, _P, _S]),
    "al_axpy": (ct.c_int, [_P, _P, _P, ct.c_int64, _S]),
    "al_row_stats": (ct.c_int, [_P, ct.c_int32, ct.c_int64, _P, _P, _S]),
    "al_row_stats_partials": (ct.c_int64, [ct.c_int32, ct.c_int64]),
    "al_fx_apply": (ct.c_int, [ct.c_int, _P, _P, ct.c_int64, _P, _P, _S]),
    "al_fx_frame_shuffle": (ct.c_int, [_P, _P, ct.c_int64, ct.c_int32, ct.c_int32, _P, ct.c_int32, _S]),
    "al_pack_ragged_irs": (ct.c_int, [_P, ct.c_int32, _P, _P, ct.c_int64, ct.c_int32, _P, _S]),
    "al_resample_poly": (ct.c_int, [_P, ct.c_int32, ct.c_int64, _P, ct.c_int32, ct.c_int32, ct.c_int32, _P, ct.c_int64, ct.c_int64, _S]),
    "al_encode_frames": (ct.c_int, [_P, ct.c_int32, ct.c_int64, ct.c_int32, _P, _S]),
    "al_wrap_copy": (ct.c_int, [_P, ct.c_int64, _P, ct.c_int64, _S]),
    "al_pack_irs_f64": (ct.c_int, [_P, _P, ct.c_int64, ct.c_int32, ct.c_int32, _S]),
    "al_pack_irs_f32": (ct.c_int, [_P, _P, ct.c_int64, ct.c_int32, ct.c_int32, _S]),
    "al_noise_workspace_floats": (ct.c_int64, [ct.c_int32, ct.c_int64]),
    "al_noise_irfft": (ct.c_int, [_P, _P, _P, ct.c_int32, ct.c_int64, ct.c_float, _P, _P, _S]),
    "al_noise_irfft_seeded": (ct.c_int, [ct.c_uint64, _P, ct.c_int32, ct.c_int64, ct.c_float, _P, _P, _S]),
    "al_normal_fill": (ct.c_int, [_P, ct.c_int64, ct.c_uint64, ct.c_uint32, ct.c_float, _S]),
    "al_philox4x32_10": (ct.c_int, [ct.POINTER(ct.c_uint32), ct.POINTER(ct.c_uint32), ct.POINTER(ct.c_uint32)]),
    "al_ambience_scales": (ct.c_int, [_P, ct.c_int32, ct.c_int64, ct.c_float, ct.c_int32, _P, _S]),
    "al_axpy_rows": (ct.c_int, [_P, _P, _P, ct.c_int32, ct.c_int64, _S]),
    "al_stft_workspace_floats": (ct.c_int64, [ct.c_int64, ct.c_int32]),
    "al_stft": (ct.c_int, [_P, ct.c_int64, ct.c_int64, ct.c_int32, ct.c_int32, ct.c_int32, _P, _P, _S]),
    "al_tv_stft_mac": (ct.c_int, [_P, _P, _P, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_int32, _P, _S]),
    "al_istft_workspace_floats": (ct.c_int64, [ct.c_int32, ct.c_int32, ct.c_int32]),
    "al_istft_ola": (ct.c_int, [_P, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_int32, _P, _P, _S]),
    "al_scale_matrix_rows": (ct.c_int, [_P, ct.c_int32, ct.c_int64, _P, _S]),
}
FRAMES_F32, FRAMES_PCM16 = 0, 1
FX_GAIN, FX_INVERT, FX_REVERSE, FX_FADE, FX_CLIP, FX_TANH, FX_BITCRUSH, FX_PREEMPH, FX_DEEMPH = range(1, 10)
FADE_SHAPES = {"linear": 0, "exponential": 1, "logarithmic": 2, "quarter_sine": 3, "half_sine": 4, "none": 5}


class Library:
    """Loaded C-ABI library; ``call(name, *args)`` raises HipError on a negative status."""

    def __init__(self, path: Optional[str] = None):
        path = path or os.environ.get("AUDIBLELIGHT_HIP_LIB") or DEFAULT_LIB
        if not os.path.exists(path):
            raise RuntimeError(
                f"audiblelight_amd: HIP extension not found at {path}. Build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                f"There is no CPU fallback.")
        self.path = path
        # torch ships its own libamdhip64/libhsa-runtime64; load them first so this library binds to
        # the SAME HIP runtime instance (two runtimes in one process cannot both own the GPU).
        import torch  # noqa: F401

        self._dll = ct.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(self._dll, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
            setattr(self, "_" + name, fn)
        built = self._al_abi_version()
        if built != ABI_VERSION:
            raise RuntimeError(f"audiblelight_amd: {path} implements C-ABI version {built}, this package binds version "
                               f"{ABI_VERSION}: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")

    def last_error(self) -> str:
        return (self._al_last_error() or b"").decode()

    def call(self, name: str, *args):
        rc = getattr(self, "_" + name)(*args)
        if isinstance(rc, int) and rc < 0 and SYMBOLS[name][0] is ct.c_int:
            planner = name.startswith(("al_plan", "al_mix_plan", "al_interpolation"))   # csrc/al_plan.cpp keeps its own message
            raise HipError(f"{name} failed ({rc}): {(self._al_plan_last_error() or b'').decode() if planner else self.last_error()}")
        return rc


_default: Optional[Library] = None


def get_library() -> Library:
    global _default
    if _default is None:
        _default = Library()
    return _default


# ----------------------------------------------------------------------------- the planner alone
# csrc/al_plan.cpp is plain C++ with no device call: build() also links it into a library of its own, so that planning (tables,
# workspace sizes, block size, dispatch flags) runs on a host without ROCm, without importing torch, and before any GPU is
# touched.  A Renderer plans with ITS library (the full one exports the same symbols); everything else uses this one.
DEFAULT_PLANNER = os.path.join(_HERE, "csrc", "libaudiblelight_plan.so")
PLANNER_SYMBOLS = ("al_plan_last_error", "al_plan_abi_version", "al_choose_log2_block", "al_stft_frame_count", "al_interpolation_rows",
                   "al_interpolation_matrix", "al_plan_create", "al_plan_destroy", "al_plan_get_info", "al_plan_events",
                   "al_plan_streams", "al_plan_wtab", "al_plan_audio_offsets", "al_workspace_bytes", "al_plan_chunk",
                   "al_plan_emitter_parts", "al_plan_batch_flags", "al_plan_mixdown", "al_mix_plan_destroy", "al_mix_plan_get")


class PlannerLibrary:
    """The host-side planner behind the C ABI, loaded from the planner-only library; same ``call`` contract as ``Library``."""

    def __init__(self, path: Optional[str] = None):
        path = path or os.environ.get("AUDIBLELIGHT_PLAN_LIB") or DEFAULT_PLANNER
        if not os.path.exists(path):
            raise RuntimeError(f"audiblelight_amd: planner library not found at {path}. Build it with "
                               f"`python -c 'import __graft_entry__ as g; g.build()'`.")
        self.path = path
        self._dll = ct.CDLL(path)
        for name in PLANNER_SYMBOLS:
            fn = getattr(self._dll, name)
            fn.restype, fn.argtypes = SYMBOLS[name]
            setattr(self, "_" + name, fn)
        built = self._al_plan_abi_version()
        if built != ABI_VERSION:
            raise RuntimeError(f"audiblelight_amd: {path} implements C-ABI version {built}, this package binds version {ABI_VERSION}: rebuild it")

    def call(self, name: str, *args):
        rc = getattr(self, "_" + name)(*args)
        if isinstance(rc, int) and rc < 0 and SYMBOLS[name][0] is ct.c_int:
            raise HipError(f"{name} failed ({rc}): {(self._al_plan_last_error() or b'').decode()}")
        return rc


_planner: Optional[PlannerLibrary] = None


def get_planner():
    """The planner-only library; where it is absent (a tree built before it existed, or only AUDIBLELIGHT_HIP_LIB pointing at a
    custom / sanitizer build) the FULL library, which exports the same al_plan_* symbols.  Raises only if neither exists."""
    global _planner
    if _planner is None:
        path = os.environ.get("AUDIBLELIGHT_PLAN_LIB") or DEFAULT_PLANNER
        if os.path.exists(path) or not os.path.exists(os.environ.get("AUDIBLELIGHT_HIP_LIB") or DEFAULT_LIB):
            _planner = PlannerLibrary(path)      # (raises with the build hint when neither library is there)
        else:
            _planner = get_library()
    return _planner
