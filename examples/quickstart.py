#!/usr/bin/env python
"""Render one small scene on the MI355X with the AudibleLight-style API and write a WAV per microphone.

    python examples/quickstart.py [output_dir]

Clips and IRs are synthetic here; with the real AudibleLight package installed, call
``audiblelight_amd.dropin.install()`` instead and keep using ``audiblelight.core.Scene`` unchanged.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from audiblelight_amd import ambience, augmentation, core  # noqa: E402


def main(out_dir="quickstart_out"):
    rng = np.random.default_rng(0)
    sr, n_caps, ir_len = 24000, 4, 12000
    decay = np.exp(-np.arange(ir_len) / (ir_len / 6.9))
    # (capsules, emitters, samples): one static source and one source moving over 4 waypoints
    irs = rng.standard_normal((n_caps, 5, ir_len)) * decay
    scene = core.Scene(duration=10.0, state=core.StaticIRState({"mic000": irs}), sample_rate=sr, ref_db=-65)
    scene.add_event(core.Event("static", rng.standard_normal(2 * sr).astype(np.float32), sr, snr=15, scene_start=1.0,
                               augmentations=[augmentation.Gain(sr, gain_db=-3.0), augmentation.Fade(sr, 0.1, 0.3, "linear", "half_sine")]))
    scene.add_event(core.Event("moving", rng.standard_normal(4 * sr).astype(np.float32), sr, snr=20, scene_start=4.0, n_emitters=4))
    scene.add_ambience(ambience.Ambience(channels=n_caps, duration=10.0, alias="pink", noise="pink", ref_db=-70, sample_rate=sr))
    audio = scene.generate(output_dir=out_dir, metadata_dcase=False)   # events from bare arrays carry no DCASE class indices
    for mic, buf in audio.items():
        print(f"{mic}: {buf.shape} float32, peak {np.abs(buf).max():.3e} -> {out_dir}/audio_out_{mic}.wav")
    json_path = os.path.join(out_dir, "scene.json")
    scene.to_json(json_path)
    again = core.Scene.from_json(json_path, clips={a: e._raw for a, e in scene.events.items()}, irs={"mic000": irs})
    assert np.array_equal(again.generate()["mic000"], audio["mic000"])
    print("re-rendered from", json_path, "bit-identically")


if __name__ == "__main__":
    main(*sys.argv[1:])
