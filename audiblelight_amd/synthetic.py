"""Synthetic workloads of BASELINE.json / SURVEY.md section 8d (host-side input generation only)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from .plan import EventSpec

CONFIGS = {
    # name: sample_rate, capsules, events, emitters per event, ir_len, clip_len, scene seconds
    "cfg1": dict(sr=24000, C=4, E=4, N=1, Lir=12000, La=48000, T=10.0),
    "cfg2": dict(sr=48000, C=32, E=64, N=1, Lir=96000, La=192000, T=60.0),
    "cfg3": dict(sr=48000, C=32, E=16, N=32, Lir=96000, La=372000, T=60.0),
    "cfg4": dict(sr=48000, C=32, E=32, N=1, Lir=48000, La=192000, T=30.0),
    "cfg5": dict(sr=48000, C=64, E=128, N=1, Lir=192000, La=192000, T=60.0, ambience="white", fx=True),
}


@dataclass
class SyntheticScene:
    name: str
    sr: int
    duration: float
    n_capsules: int
    ir_len: int
    clips: List[np.ndarray]     # float32, peak-normalised
    irs: np.ndarray             # (C, sum N, Lir) float32
    specs: List[EventSpec]
    starts: List[float]
    irs_dev: object = None        # bench only: the IR tensor drawn ON THE DEVICE (flat float32, same law), then `irs` is None
    ir_shape: tuple = ()          # (C, sum N, Lir) whichever side holds the tensor
    ambience_beta: object = None  # noise colour of the scene ambience (cfg5: "white"), None = no ambience
    gain_db: object = None        # cfg5: per-event Gain(gain_db) of the [Gain, Invert] chain; `clips` are then RAW clips

    def sources(self):
        """What the renderer is given per event: finished clips, or (cfg5) the raw clip plus the folded scalar of its
        [Gain(gain_db), Invert] chain with the peak normalisation left to the device (engine.ClipSource)."""
        from .engine import ClipSource

        if self.gain_db is None:
            return self.clips
        return [ClipSource(host=c, n=len(c), prescale=-float(np.float32(10.0 ** (g / 20.0))), normalize=True)
                for c, g in zip(self.clips, self.gain_db)]

    def describe(self) -> str:
        sp = self.specs[0]
        kind = f"{sp.n_emitters}-IR moving" if sp.is_moving else "static"
        return (f"{self.name}: 1 scene/GPU/step, {self.n_capsules} capsules, {len(self.specs)} {kind} events, "
                f"{self.ir_len / self.sr:g} s RIR, {len(self.clips[0]) / self.sr:g} s clips, {self.duration:g} s scene @ "
                f"{self.sr} Hz" + (f", {self.ambience_beta} ambience" if self.ambience_beta is not None else "")
                + (", [Gain, Invert] + peak normalisation per event evaluated on the device and folded into the clip spectra"
                   if self.gain_db is not None else ""))

    @property
    def ends(self):
        return [s + len(c) / self.sr for s, c in zip(self.starts, self.clips)]

    def algorithmic_bytes(self) -> int:
        """Inputs read once + scene.audio written once (SURVEY 8d "scene-only contract")."""
        audio = sum(len(c) for c in self.clips) * 4
        n_ir = int(np.prod(self.ir_shape)) if self.irs is None else self.irs.size
        return audio + n_ir * 4 + self.n_capsules * round(self.duration * self.sr) * 4

    def algorithmic_bytes_full_api(self) -> int:
        """SURVEY 8d "full-API contract": the scene-only bytes plus every event's (C, La) ``event.spatial_audio`` written once --
        the API exposes it and the kernels do materialise it (un-scaled, in HBM; read again by the mixdown)."""
        return self.algorithmic_bytes() + sum(len(c) for c in self.clips) * self.n_capsules * 4


def make_scene(name: str = "cfg2", scene_index: int = 0, scale: float = 1.0, torch_device=None, **override) -> SyntheticScene:
    """White-noise clips + exponentially decaying random IRs with a unit direct tap.

    ``torch_device``: draw the IR tensor on that device instead (torch.randn, same law: N(0,1) x exp(-t / tau) plus a unit tap
    at a random early sample per row); for timing legs whose 6.3 GB of host draws would take longer than the measurement.
    The parity tests always use the host draws."""
    cfg = dict(CONFIGS[name])
    cfg.update(override)
    sr, C, E, N = cfg["sr"], cfg["C"], cfg["E"], cfg["N"]
    Lir, La, T = int(cfg["Lir"] * scale), int(cfg["La"] * scale), cfg["T"] * scale
    rng = np.random.default_rng(1234 + scene_index)
    decay = np.exp(-np.arange(Lir, dtype=np.float32) / np.float32(Lir / 6.9))
    irs = np.empty((C, E * N, Lir), dtype=np.float32) if torch_device is None else None
    irs_dev = None
    if torch_device is not None:
        import torch

        gen = torch.Generator(device=torch_device)
        gen.manual_seed(1234 + scene_index)
        irs_dev = torch.randn((C * E * N, Lir), generator=gen, device=torch_device, dtype=torch.float32)
        irs_dev *= torch.from_numpy(decay).to(torch_device)[None, :]
        taps = torch.randint(48, min(960, Lir), (C * E * N,), generator=gen, device=torch_device)
        irs_dev[torch.arange(C * E * N, device=torch_device), taps] += 1.0
        irs_dev = irs_dev.reshape(-1)
    clips, specs, starts, gains_db = [], [], [], []
    for e in range(E):
        a = rng.standard_normal(La, dtype=np.float32)
        if not cfg.get("fx"):   # finished clip: peak-normalised as Event.load_audio leaves it (event.py:535-536)
            a = a / np.max(np.abs(a) + np.finfo(np.float32).tiny)
        clips.append(a.astype(np.float32))
        for n in range(N if irs is not None else 0):
            h = rng.standard_normal((C, Lir), dtype=np.float32) * decay
            h[np.arange(C), rng.integers(48, min(960, Lir), size=C)] += 1.0
            irs[:, e * N + n, :] = h
        if cfg.get("fx"):   # raw clip + [Gain(gain_db ~ U(-10, 10)), Invert]: see SyntheticScene.sources
            gains_db.append(float(rng.uniform(-10, 10)))
        specs.append(EventSpec(n_samples=La, n_emitters=N, snr=float(rng.uniform(5, 30)), emitter0=e * N,
                               is_moving=N > 1, duration=La / sr, ref_db=-65.0))
        starts.append(float(rng.uniform(0, max(T - La / sr, 0.0))))
    return SyntheticScene(name=name, sr=sr, duration=T, n_capsules=C, ir_len=Lir, clips=clips, irs=irs, specs=specs,
                          starts=starts, irs_dev=irs_dev, ir_shape=(C, E * N, Lir), ambience_beta=cfg.get("ambience"),
                          gain_db=gains_db if cfg.get("fx") else None)
