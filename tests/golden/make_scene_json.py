#!/usr/bin/env python
"""Reference-format scene metadata + the reference's render of it (this container only).

Builds a small scene from duck-typed objects (the attribute list of SURVEY.md 8a A15), serialises it with the REAL
reference's own ``Scene.to_dict`` / ``Event.to_dict`` / ``Ambience.to_dict`` / ``MicArray.to_dict`` code (called on
those objects), renders it with the real ``render_audio_for_all_scene_events`` + ``generate_scene_audio_from_events``
and stores
    tests/golden/reference_scene.json        the metadata exactly as the reference lays it out (core.py:2106-2130,
                                             event.py:568-620, ambience.py:219-233, worldstate.py:2330-2356)
    tests/golden/reference_scene_arrays.npz  the arrays the JSON does not carry (decoded clips, IR tensors) and the
                                             reference's outputs (scene.audio, per-event spatial audio)
Only data is stored; nothing of the reference's source travels.

    python tests/golden/make_scene_json.py
"""
import importlib.metadata
import json
import os
import sys
import types
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    syn, amb = mg._import_reference()
    import audiblelight.core as ref_core
    import audiblelight.event as ref_event
    import audiblelight.micarrays as ref_mic

    ver = importlib.metadata.version
    importlib.metadata.version = lambda n: "0.0.0-standin" if n == "rlr_audio_propagation" else ver(n)
    ref_core.version = importlib.metadata.version

    rng = np.random.default_rng(20261004)
    sr, dur = 8000, 3.0
    mics = OrderedDict(mic000=ref_mic.AmbeoVR(), mic001=ref_mic.AmbeoVR())   # equal capsule counts: the reference requires the
    # ambience to match every microphone's scene shape (synthesize.py:343-347)
    for m in mics.values():   # absolute coordinates are set by the WorldState when a microphone is placed
        try:
            m.set_absolute_coordinates(np.array([1.0, 2.0, 1.5]))
        except Exception:  # noqa: BLE001
            pass
    caps = {k: m.n_capsules for k, m in mics.items()}

    def emitter(xyz):
        return types.SimpleNamespace(coordinates_absolute=np.asarray(xyz, dtype=float),
                                     coordinates_relative_polar=OrderedDict((k, np.array([[30.0, 10.0, 2.0]])) for k in mics))

    spec = [  # alias, n_audio, n_emitters, scene_start, snr, moving, augmentations (reference to_dict layout), dry
        ("event000", 9000, 1, 0.20, 12.0, False, [], None),
        ("event001", 7000, 1, 0.90, 20.0, False, [dict(name="Gain", sample_rate=sr, gain_db=-3.0), dict(name="Invert", sample_rate=sr)], None),
        ("event002", 8000, 4, 1.50, 9.0, True, [], None),
        ("event003", 6000, 1, 2.60, 15.0, False, [], (0, [2, 20])),    # runs past the scene end + dry path
    ]
    events, clips = OrderedDict(), {}
    for alias, n, ne, st, snr, mv, augs, dry in spec:
        raw = rng.standard_normal(n).astype(np.float32) * np.float32(rng.uniform(0.2, 0.9))
        clips[alias] = raw
        a = raw.copy()
        for d in augs:   # what the reference's chain computes for these two pure-scalar FX (augmentation.py:1105-1136,1557-1580)
            a = a * np.float32(10.0 ** (d["gain_db"] / 20.0)) if d["name"] == "Gain" else -a
        a = a / np.max(np.abs(a) + np.finfo(np.float32).tiny)
        ev = mg.FakeEvent(alias, a.astype(np.float32), ne, snr, sr, scene_start=st, is_moving=mv,
                          ref_ir_channel=dry[0] if dry else None, direct_path_time_ms=dry[1] if dry else None)
        ev.has_emitters, ev.emitters = True, [emitter(rng.uniform(0, 3, 3)) for _ in range(ne)]
        ev.filename, ev.filepath = f"{alias}.wav", f"/data/fg/{alias}.wav"
        ev.class_id, ev.class_label = 3, "telephone"
        ev.event_start, ev.event_end = 0.5, 0.5 + ev.duration
        ev.image_filepath, ev.shape = None, None
        ev.spatial_resolution, ev.spatial_velocity = (4.0, 1.0) if mv else (None, None)
        ev.augmentations = [types.SimpleNamespace(to_dict=lambda d=d: dict(d)) for d in augs]
        ev.to_dict = types.MethodType(ref_event.Event.to_dict, ev)   # the reference's own Event.to_dict on this object
        events[alias] = ev
    irs = {k: np.concatenate([mg.make_irs(rng, c, len(e), 1200) for e in events.values()], axis=1) for k, c in caps.items()}
    ambience = OrderedDict(bg=amb.Ambience(channels=max(caps.values()), duration=dur, alias="bg", noise="pink", ref_db=-60, sample_rate=sr))

    state = types.SimpleNamespace(
        name="RLR_standin", irs=irs, get_irs=lambda: irs, simulate=lambda: None, microphones=mics,
        num_emitters=sum(len(e) for e in events.values()),
        to_dict=lambda: dict(backend="RLR_standin", sample_rate=sr,
                             emitters={a: [e_.coordinates_absolute.tolist() for e_ in ev.emitters] for a, ev in events.items()},
                             microphones={k: m.to_dict() for k, m in mics.items()},
                             mesh=dict(fpath="/data/mesh.glb", bounds=[[0, 0, 0], [4, 5, 3]], centroid=[2, 2.5, 1.5])))
    scene = types.SimpleNamespace(state=state, events=events, ambience=ambience, audio={}, ref_db=-60, duration=dur,
                                  sample_rate=sr, max_overlap=3, fg_paths=[], bg_paths=[], class_mapping=None)
    meta = ref_core.Scene.to_dict(scene)                       # the reference's own serialiser (calls Event.to_dict etc.)
    syn.render_audio_for_all_scene_events(scene)
    syn.generate_scene_audio_from_events(scene)

    out = {f"clip_{k}": v for k, v in clips.items()}
    out.update({f"irs_{k}": v.astype(np.float32) for k, v in irs.items()})
    for k in caps:
        out[f"scene_{k}"] = scene.audio[k]
        for alias, ev in events.items():
            out[f"spatial_{k}_{alias}"] = ev.spatial_audio[k].astype(np.float32)
    out["dry_mic000_event003"] = events["event003"]._spatial_audio_dry["mic000"]
    with open(os.path.join(HERE, "reference_scene.json"), "w") as fh:
        json.dump(json.loads(json.dumps(meta, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))), fh, indent=1)
    path = os.path.join(HERE, "reference_scene_arrays.npz")
    np.savez_compressed(path, **out)
    print("wrote reference_scene.json", os.path.getsize(os.path.join(HERE, "reference_scene.json")) // 1024, "KiB and",
          path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
