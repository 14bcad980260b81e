"""DCASE-2024 metadata rows of a rendered Scene (reference audiblelight/synthesize.py:742-878).

Host-side bookkeeping about event POSITIONS: no samples are involved and nothing here touches the GPU.  SURVEY.md section 8 (f4) leaves
this step "unchanged on host", outside the accelerated path; it lives in a module of its own (not in ``synthesize``, the hot-path
module) so that ``Scene.generate(metadata_dcase=True)`` and ``batch.render_dataset`` can still write the per-microphone CSVs a dataset
job expects.  ``audiblelight_amd.synthesize.generate_dcase2024_metadata`` stays importable under the reference's name (a lazy
forwarder): pandas is only imported when the function is called.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

DCASE_2024_COLUMNS = ["frame_number", "active_class_index", "source_number_index", "azimuth", "elevation", "distance"]


def _polar_of(event, mic: str) -> np.ndarray:
    """(n_emitters, 3) azimuth / elevation (degrees) and distance (metres) of an event's emitters relative to ``mic``: the
    reference's ``Emitter.coordinates_relative_polar[mic]`` objects, or the ``emitters_relative`` block of its metadata."""
    emitters = getattr(event, "emitters", None)
    if emitters and hasattr(emitters[0], "coordinates_relative_polar"):
        return np.vstack([np.asarray(e.coordinates_relative_polar[mic], dtype=float).reshape(-1, 3) for e in emitters])
    rel = getattr(event, "emitters_relative", None) or getattr(event, "metadata", {}).get("emitters_relative")
    if not rel or mic not in rel:
        raise ValueError(f"Event {getattr(event, 'alias', '?')} carries no emitter positions relative to {mic}: "
                         "DCASE metadata needs them (Event.metadata['emitters_relative'])")
    return np.asarray(rel[mic], dtype=float).reshape(-1, 3)


def generate_dcase2024_metadata(scene, temporal_resolution=0.1):
    """DCASE-2024 rows per microphone (reference synthesize.py:742-878): one row per active 100 ms frame and event --
    frame, class index, source index (counted per class in order of appearance; events sharing an audio file share it),
    rounded azimuth / elevation in degrees and distance in centimetres, linearly interpolated over the emitters of a moving
    event; sorted by (frame, class, source), indexed by frame.  Pure host bookkeeping (pandas), no samples involved."""
    import pandas as pd

    frames = np.round(np.arange(0, scene.duration + temporal_resolution, temporal_resolution), 1)
    microphones = list(scene.state.microphones.keys())
    rows = {mic: [] for mic in microphones}
    events = scene.get_events() if hasattr(scene, "get_events") else list(scene.events.values())
    next_source: Dict[int, int] = {}
    source_of_file: Dict[str, int] = {}

    def frame_of(t: float) -> int:
        return int(np.where(frames == round(t, 1))[0][0])

    for event in sorted(events, key=lambda e: e.scene_start):
        span = np.arange(frame_of(max(event.scene_start, 0.0)), frame_of(min(event.scene_end, scene.duration)) + 1)
        if not isinstance(event.class_id, int):
            raise ValueError("Can't convert Event to DCASE format without valid DCASE class indices")
        fname = getattr(event, "filename", None)
        if fname not in source_of_file:
            source_of_file[fname] = next_source.get(event.class_id, 0)
            next_source[event.class_id] = next_source.get(event.class_id, 0) + 1
        source = source_of_file[fname]
        for mic in microphones:
            polar = _polar_of(event, mic)
            if not event.is_moving:
                track = np.repeat(polar[:1], len(span), axis=0)
            else:
                times = frames[span]
                knots = np.linspace(times.min(), times.max(), num=len(polar))
                track = np.stack([np.interp(times, knots, polar[:, d]) for d in range(3)], axis=1)
            for idx, (az, el, dist) in zip(span, track):
                rows[mic].append([int(idx), event.class_id, source, round(az), round(el), round(dist * 100)])
    return {mic: pd.DataFrame(data, columns=DCASE_2024_COLUMNS)
            .sort_values(["frame_number", "active_class_index", "source_number_index"]).set_index("frame_number")
            for mic, data in rows.items()}
