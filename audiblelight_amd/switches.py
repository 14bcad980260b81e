"""A/B and debugging switches of the package, read from the environment ONCE.

The dispatch policy proper (which layout, which accumulate) lives behind the C ABI (``al_plan_batch_flags``,
csrc/al_plan.cpp).  What is left here are switches that force OTHER code paths than that policy -- for A/B
measurements and so that the parity tests can reach every kernel in the library -- plus a few host-side transfer
choices.  They are parsed when first asked for (``current()``), never on the per-scene path; a process that
changes the environment afterwards calls ``reload()`` (the test suite does, tests/conftest.py::set_switch).
``non_default()`` is what bench.py records in ``config.switches``, so a stray variable shows up in the result line.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, fields
from typing import Optional


def _flag(name: str):
    v = os.environ.get(name)
    return None if v is None else v == "1"


@dataclass(frozen=True)
class Switches:
    split: Optional[bool] = None          # AL_SPLIT=0/1: force the one-transform / split-layout kernels (None: library policy)
    quad16: bool = True                   # AL_QUAD16=0: B = 16384 without the quad-tile kernels (split or one-transform instead)
    static_mac: bool = True               # AL_STATIC_MAC=0: one-emitter events through the tile accumulate
    static_mac_max_p: Optional[int] = None   # AL_STATIC_MAC_MAX_P: capsule-loop accumulate only up to that many partitions
    trim_partitions: bool = True          # AL_TRIM_PARTITIONS=0: transform IR partitions no kept block hears (al_batch.emitter_parts off)
    extra_flags: int = 0                  # AL_EXTRA_FLAGS: A/B bits of al_batch.flags, masked to those that do not change results
    f64_upload: str = "host"              # AL_F64_UPLOAD=device: float64 IRs cast on the device instead of by host threads
    convert_threads: Optional[int] = None    # AL_CONVERT_THREADS: host threads of that cast
    beside_min_bytes: int = 8 << 20       # AL_BESIDE_MIN_BYTES: smallest IR tensor uploaded from the helper thread
    workspace_gb: Optional[float] = None  # AL_WORKSPACE_GB: spectra workspace budget (None: 40 % of the free HBM)
    d2h: str = "dma"                      # AL_D2H=kernel: batch driver stores straight into page-locked host memory
    h2d: str = "blocking"                 # AL_H2D=async: batch driver page-locks the caller's arrays in place
    ambience_rng: str = "host"            # AL_AMBIENCE_RNG=device: Ambience draws its noise on the GPU by default

    @classmethod
    def from_env(cls) -> "Switches":
        env = os.environ.get
        mode = env("AL_AMBIENCE_RNG", "host")
        if mode not in ("host", "device"):
            raise ValueError(f"AL_AMBIENCE_RNG must be 'host' or 'device', got {mode!r}")
        return cls(
            split=_flag("AL_SPLIT"),
            quad16=env("AL_QUAD16", "1") == "1",
            static_mac=env("AL_STATIC_MAC", "1") == "1",
            static_mac_max_p=int(env("AL_STATIC_MAC_MAX_P")) if env("AL_STATIC_MAC_MAX_P") else None,
            trim_partitions=env("AL_TRIM_PARTITIONS", "1") == "1",
            extra_flags=int(env("AL_EXTRA_FLAGS", "0")),
            f64_upload=env("AL_F64_UPLOAD", "host"),
            convert_threads=int(env("AL_CONVERT_THREADS")) if env("AL_CONVERT_THREADS") else None,
            beside_min_bytes=int(env("AL_BESIDE_MIN_BYTES", 8 << 20)),
            workspace_gb=float(env("AL_WORKSPACE_GB")) if env("AL_WORKSPACE_GB") else None,
            d2h=env("AL_D2H", "dma"), h2d=env("AL_H2D", "blocking"), ambience_rng=mode)

    def non_default(self) -> dict:
        """{field: value} of every switch that differs from the default (empty: the library's own policy everywhere)."""
        base = Switches()
        return {f.name: getattr(self, f.name) for f in fields(self) if getattr(self, f.name) != getattr(base, f.name)}

    @property
    def forces_dispatch(self) -> bool:
        """True when a switch overrides the kernel choice of ``al_plan_batch_flags``."""
        return self.split is not None or not self.quad16 or not self.static_mac or self.static_mac_max_p is not None


_current: Optional[Switches] = None


def current() -> Switches:
    global _current
    if _current is None:
        _current = Switches.from_env()
    return _current


def set_env(name: str, value) -> Switches:
    """Set (value None: remove) one AL_* variable in this process and parse the switches again: for measurement drivers and tests
    that flip a switch between two renders (profiles/tools/*.py, tests/conftest.py::set_switch)."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    return reload()


def reload() -> Switches:
    """Parse the environment again (tests that flip a switch between two renders)."""
    global _current
    _current = Switches.from_env()
    return _current
