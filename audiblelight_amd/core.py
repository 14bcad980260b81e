"""Minimal Scene / Event / WorldState stand-ins with the attribute surface the synthesis path uses.

The reference's ``Scene`` (audiblelight/core.py), ``Event`` (audiblelight/event.py) and ``WorldState``
(audiblelight/worldstate.py) do placement, ray tracing and file decoding, which are out of scope
here.  These classes carry exactly the attributes listed in SURVEY.md 8a A15,
so a scene can be described from arrays already in memory (clips + IR tensors) and rendered with
``Scene.generate()``; objects of the real AudibleLight classes work with the same functions.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import config
from .utils import LazyAudioDict, tiny, valid_audio


class Event:
    """An audio event: a mono clip, its emitters (IR columns), timing and level (event.py:31-191)."""

    def __init__(self, alias: str, audio: np.ndarray, sample_rate: int = config.SAMPLE_RATE, snr: float = 15.0,
                 scene_start: float = 0.0, n_emitters: int = 1, is_moving: Optional[bool] = None,
                 augmentations: Sequence = (), ref_ir_channel: Optional[int] = None,
                 direct_path_time_ms: Optional[Sequence[float]] = None, class_label: Optional[str] = None):
        audio = np.asarray(audio)
        if audio.ndim != 1:
            raise ValueError("Event audio must be mono (1-D)")
        self.alias = alias
        self.sample_rate = int(sample_rate)
        self._raw = np.ascontiguousarray(audio, dtype=np.float32)
        self.snr = float(snr)
        self.scene_start = float(scene_start)
        self.duration = len(self._raw) / self.sample_rate
        self.scene_end = self.scene_start + self.duration
        self.n_emitters = int(n_emitters)
        self.is_moving = bool(self.n_emitters > 1 if is_moving is None else is_moving)
        self.augmentations: List = list(augmentations)
        self.ref_ir_channel = ref_ir_channel
        self.direct_path_time_ms = direct_path_time_ms
        self.class_label = class_label
        self.audio: Optional[np.ndarray] = None
        self.spatial_audio = LazyAudioDict()
        self._spatial_audio_padded = LazyAudioDict()
        self._spatial_audio_dry: Dict[str, np.ndarray] = OrderedDict()
        self._spatial_audio_dry_padded: Dict[str, np.ndarray] = OrderedDict()

    def __len__(self) -> int:
        return self.n_emitters

    @property
    def is_audio_loaded(self) -> bool:
        return self.audio is not None

    def register_augmentations(self, augmentations) -> None:
        """Add FX and drop every cached render (event.py:739-782)."""
        self.augmentations.extend(augmentations if isinstance(augmentations, (list, tuple)) else [augmentations])
        self.clear_audio()

    def clear_augmentations(self) -> None:
        self.augmentations = []
        self.clear_audio()

    def clear_audio(self) -> None:
        self.audio = None
        self._last_chain = None
        self.spatial_audio = LazyAudioDict()
        self._spatial_audio_padded = LazyAudioDict()
        self._spatial_audio_dry = OrderedDict()
        self._spatial_audio_dry_padded = OrderedDict()

    def _device_chain(self, normalize: bool):
        """Raw clip -> HBM once, the whole FX chain and the peak normalisation there (augmentation.run_chain)."""
        from . import augmentation, synthesize

        device_fx = all(hasattr(a, "process_device") for a in self.augmentations)
        if not device_fx:   # foreign callables (e.g. host pedalboard FX of the reference): run them where they live
            out = self._raw.copy()
            for aug in self.augmentations:
                out = aug(out)
            clip = augmentation.DeviceClip(synthesize.get_renderer(), out)
            return augmentation.run_chain(clip, [], normalize)
        clip = augmentation.DeviceClip(synthesize.get_renderer(), self._raw)
        return augmentation.run_chain(clip, self.augmentations, normalize)

    def load_audio(self, ignore_cache: Optional[bool] = False, normalize: Optional[bool] = True) -> np.ndarray:
        """Clip after the FX chain and peak normalisation (event.py:496-539); cached in ``self.audio``.
        One upload, the chain on the device, one download (no intermediate copies between FX)."""
        if self.is_audio_loaded and not ignore_cache:
            return self.audio
        clip = self._device_chain(bool(normalize))
        out = clip.host()
        valid_audio(out)
        self.audio = out
        self._last_chain = clip
        return self.audio

    def clip_source(self, ignore_cache: Optional[bool] = False):
        """What the renderer needs of this event's clip WITHOUT bringing samples back to the host
        (engine.ClipSource): a chain of pure scalars (Gain, Invert) + peak normalisation becomes the raw clip plus
        one device-evaluated scalar; any other chain runs on the device and is handed over in HBM; a clip somebody
        already loaded to the host (``self.audio``) is used as it is."""
        from . import augmentation, engine

        if self.is_audio_loaded and not ignore_cache:
            return engine.ClipSource(host=np.ascontiguousarray(self.audio, dtype=np.float32), n=len(self.audio))
        folded = augmentation.fold_scalars(self.augmentations)
        if folded is not None:
            return engine.ClipSource(host=self._raw, n=len(self._raw), prescale=folded, normalize=True)
        clip = self._device_chain(True)
        self._last_chain = clip
        return engine.ClipSource(device=clip.buf, n=clip.n)

    def to_dict(self) -> dict:
        return dict(alias=self.alias, sample_rate=self.sample_rate, snr=self.snr, scene_start=self.scene_start,
                    scene_end=self.scene_end, duration=self.duration, n_emitters=self.n_emitters,
                    is_moving=self.is_moving, class_label=self.class_label,
                    augmentations=[a.to_dict() for a in self.augmentations if hasattr(a, "to_dict")])


class MicArray:
    """Capsule count holder (micarrays.py:36-162): only ``n_capsules`` / ``n_listeners`` matter to the path."""

    def __init__(self, alias: str, n_capsules: int):
        self.alias, self.n_capsules, self.n_listeners = alias, int(n_capsules), int(n_capsules)


class StaticIRState:
    """WorldState stand-in holding precomputed IRs: {mic: (C, N_emitters_total, L)} (worldstate.py:360-370)."""

    name = "static"

    def __init__(self, irs: Dict[str, np.ndarray]):
        self._irs = OrderedDict((k, np.asarray(v)) for k, v in irs.items())
        self.microphones = OrderedDict((k, MicArray(k, v.shape[0])) for k, v in self._irs.items())

    @property
    def irs(self):
        return self._irs

    @property
    def num_emitters(self) -> int:
        return next(iter(self._irs.values())).shape[1] if self._irs else 0

    def get_irs(self):
        return self._irs

    def simulate(self) -> None:  # IRs are given, nothing to trace
        return None


class Scene:
    """Container with the attributes the synthesis functions read (core.py:131-251) and ``generate``."""

    def __init__(self, duration: float, state: StaticIRState, sample_rate: int = config.SAMPLE_RATE,
                 ref_db: float = config.DEFAULT_REF_DB):
        self.duration, self.state, self.sample_rate, self.ref_db = float(duration), state, int(sample_rate), ref_db
        self.events: "OrderedDict[str, Event]" = OrderedDict()
        self.ambience: "OrderedDict[str, object]" = OrderedDict()
        self.audio: Dict[str, np.ndarray] = OrderedDict()

    def add_event(self, event: Event) -> Event:
        if event.alias in self.events:
            raise KeyError(f"Event with alias {event.alias} already exists")
        self.events[event.alias] = event
        return event

    def add_ambience(self, ambience) -> None:
        self.ambience[ambience.alias] = ambience

    # -- metadata round trip (reference core.py:2106-2243): everything except the samples themselves
    def to_dict(self) -> dict:
        return dict(duration=self.duration, sample_rate=self.sample_rate, ref_db=self.ref_db,
                    microphones={k: m.n_capsules for k, m in self.state.microphones.items()},
                    events={k: e.to_dict() for k, e in self.events.items()},
                    ambience={k: a.to_dict() for k, a in self.ambience.items() if hasattr(a, "to_dict")})

    def to_json(self, path: str) -> None:
        import json

        with open(path, "w") as fh:
            json.dump(self.to_dict(), fh, indent=2)

    @classmethod
    def from_dict(cls, d: dict, clips: Dict[str, np.ndarray], irs: Dict[str, np.ndarray]) -> "Scene":
        """Rebuild a scene from ``to_dict`` metadata plus the arrays it does not store: ``clips[event alias]`` (raw
        mono audio) and ``irs[mic alias]`` ((C, N_total, L) tensors)."""
        from . import ambience as amb_mod, augmentation as aug_mod

        scene = cls(d["duration"], StaticIRState(irs), sample_rate=d["sample_rate"], ref_db=d["ref_db"])
        for alias, ed in d["events"].items():
            fx = [aug_mod.Augmentation.from_dict(a) for a in ed.get("augmentations", [])]
            scene.add_event(Event(alias, clips[alias], ed["sample_rate"], snr=ed["snr"], scene_start=ed["scene_start"],
                                  n_emitters=ed["n_emitters"], is_moving=ed["is_moving"], augmentations=fx,
                                  class_label=ed.get("class_label")))
        for alias, ad in d.get("ambience", {}).items():
            scene.add_ambience(amb_mod.Ambience.from_dict(ad))
        return scene

    @classmethod
    def from_json(cls, path: str, clips: Dict[str, np.ndarray], irs: Dict[str, np.ndarray]) -> "Scene":
        import json

        with open(path) as fh:
            return cls.from_dict(json.load(fh), clips, irs)

    def generate(self, output_dir=None, audio: bool = True, metadata_json: bool = True, metadata_dcase: bool = False,
                 audio_fname: str = "audio_out", metadata_fname: str = "metadata_out", video: bool = False,
                 video_fname: str = "video_out") -> Dict[str, np.ndarray]:
        """Render every event and mix the scene, with the reference's argument list (core.py:1789-1874).

        ``audio``: render on the GPU and, when ``output_dir`` is given, write float32 WAV files
        ``<audio_fname>_<mic>.wav`` ((T, C) interleaved like soundfile.write(audio.T), core.py:1840-1847).
        ``metadata_json``: write ``<metadata_fname>.json`` (``to_dict``) when ``output_dir`` is given.
        ``metadata_dcase`` / ``video`` belong to host-side subsystems that are out of scope here; asking
        for them raises instead of silently skipping.
        """
        if metadata_dcase or video:
            raise NotImplementedError("DCASE metadata and video output are host-side features of the reference "
                                      "(synthesize.py:742-878, core.py:1866) and are not part of this path")
        import os

        if output_dir is not None:
            os.makedirs(output_dir, exist_ok=True)
        if audio:
            from . import synthesize

            synthesize.render_audio_for_all_scene_events(self)
            synthesize.generate_scene_audio_from_events(self)
            if output_dir is not None:
                from scipy.io import wavfile

                stem = os.path.splitext(str(audio_fname))[0]
                for mic, buf in self.audio.items():
                    wavfile.write(os.path.join(output_dir, f"{stem}_{mic}.wav"), self.sample_rate, buf.T)
        if metadata_json and output_dir is not None:
            self.to_json(os.path.join(output_dir, os.path.splitext(str(metadata_fname))[0] + ".json"))
        return self.audio
