import sys, numpy as np
sys.path.insert(0, "/root/repo")
import os
from audiblelight_amd import _hip, engine, plan as planning, switches


def setenv(name, value):     # the package parses its AL_* switches once per process (audiblelight_amd/switches.py)
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    switches.reload()


from tests import hostemu
_hip._default = _hip.Library(sys.argv[1])
_hip._planner = _hip._default                  # the planner (csrc/al_plan.cpp) runs from the sanitized build too
r = engine.Renderer(lib=_hip._default, memory=hostemu.NumpyMemory())
rng = np.random.default_rng(0)
C, L, sr = 3, 700, 8000
specs, clips, irs, col = [], [], [], 0
for n, ne in ((1501, 1), (2100, 3), (333, 0), (1025, 1)):
    clips.append(rng.standard_normal(n).astype(np.float32))
    irs.append(rng.standard_normal((C, ne, L)).astype(np.float32))
    specs.append(planning.EventSpec(n_samples=n, n_emitters=ne, snr=10.0, emitter0=col, is_moving=ne > 1, duration=n / sr))
    col += ne
pl = planning.plan_batch(specs, C, L, sr, log2_block=10)
res = r.render(pl, clips, np.concatenate(irs, axis=1), chunk_events=2)
mix = planning.plan_mixdown([0.0, 0.1, 0.2, 0.05], [0.3, 0.4, 0.25, 0.2], [len(c) for c in clips], [C] * 4, pl.events["out_off"], [0, 1, 2, 3], 0.5, sr, C)
scene = r.mem.download(r.mixdown(mix, res))
print("asan run ok", float(np.abs(scene[: C * mix.n_samples]).sum()), res.scales())
# round 2 kernels: the LDS-staged capsule loop (clip of more than 24 blocks), its two-unit form (13 partitions, the 14th a
# zero row), rows at odd offsets (shifted-pair stores), the split spectra layout, the frame encoder's 16-byte store paths
import ctypes as ct, os
for lb, split in ((10, "0"), (11, "1")):
    setenv("AL_SPLIT", split)
    B = 1 << lb
    La, Lir, C = 26 * B + 1, 13 * B - 5, 2
    clips = [rng.standard_normal(La - 2 * e).astype(np.float32) for e in range(2)]
    irs = (rng.standard_normal((C, 2, Lir)) * np.exp(-np.arange(Lir) / (Lir / 5.0))).astype(np.float32)
    specs = [planning.EventSpec(n_samples=len(c), n_emitters=1, snr=10.0, emitter0=e) for e, c in enumerate(clips)]
    pl = planning.plan_batch(specs, C, Lir, sr, log2_block=lb)
    batch = r.prepare(pl, clips, irs)
    s_code, m_code = ct.c_int32(), ct.c_int32()
    r.lib.call("al_spectral_mac_variant", ct.byref(batch.descs[0]), ct.byref(s_code), ct.byref(m_code))
    res = batch.run()
    res.check_finite()
    print("asan run ok: block", B, "split", split, "accumulate code", s_code.value, float(np.abs(res.spatial_audio(1)).sum()))
    for P_short in (3,):   # several workgroups per (event, bin tile) with few partitions: the one-unit LDS kernel
        pl2 = planning.plan_batch(specs, C, P_short * B - 7, sr, log2_block=lb)
        res2 = r.prepare(pl2, clips, irs[:, :, : P_short * B - 7]).run()
        res2.check_finite()
for C_enc, fmt in ((8, _hip.FRAMES_PCM16), (4, _hip.FRAMES_F32), (5, _hip.FRAMES_PCM16)):
    T = 333
    scene_in = r.mem.upload(rng.standard_normal(C_enc * T).astype(np.float32))
    out = r.mem.zeros(C_enc * T, np.int16 if fmt == _hip.FRAMES_PCM16 else np.float32)
    r.lib.call("al_encode_frames", r.mem.ptr(scene_in), C_enc, T, fmt, r.mem.ptr(out), r.mem.stream())
print("asan run ok: round-2 kernels")
# round 3 kernels: the LDS-DMA capsule loop ran above (13 partitions -> accumulate code ...04; the DMA pieces are plain copies
# under emulation); here three units of 7 with the window ends in LDS, the device-side normal draws, the seeded noise
# transform (even and odd length), the per-channel ambience multipliers and the row-wise axpy
setenv("AL_SPLIT", "0")
B, C = 1024, 2
clips = [rng.standard_normal(14 * B + 3).astype(np.float32)]
irs = (rng.standard_normal((C, 1, 20 * B - 9)) * np.exp(-np.arange(20 * B - 9) / (4.0 * B))).astype(np.float32)
pl = planning.plan_batch([planning.EventSpec(n_samples=len(clips[0]), n_emitters=1, snr=10.0)], C, irs.shape[2], sr, log2_block=10)
batch = r.prepare(pl, clips, irs)
r.lib.call("al_spectral_mac_variant", ct.byref(batch.descs[0]), ct.byref(s_code), ct.byref(m_code))
res = batch.run()
res.check_finite()
print("asan run ok: 20 partitions, accumulate code", s_code.value)
from audiblelight_amd import ambience as amb, synthesize as syn
syn.set_renderer(r)
for beta, n in ((0, 1003), (1, 1000), (1, 1001)):
    x = amb.powerlaw_psd_gaussian(beta, (3, n), seed=5, rng="device")
    assert np.isfinite(x).all()
a = amb.Ambience(3, 0.125, alias="a", noise="pink", sample_rate=sr, rng="device")
noise, scales = a.noise_and_scales_device(r, (3, 1000))
scene_buf = r.mem.zeros(3 * 1000)
r.lib.call("al_axpy_rows", r.mem.ptr(scene_buf), r.mem.ptr(noise), r.mem.ptr(scales), 3, 1000, r.mem.stream())
assert np.isfinite(r.mem.download(scene_buf)).all()
syn.set_renderer(None)
print("asan run ok: round-3 kernels")

# round 4: the planner behind the C ABI ran for every batch above (al_plan_create / al_plan_chunk / al_plan_emitter_parts /
# al_plan_mixdown / al_plan_batch_flags from the sanitized library)
setenv("AL_SPLIT", None)     # back to the library's own layout policy
# the quad-tile transforms at B = 16384 (csrc/al_quad16.h): a run of five IR partitions with a ragged last one (the prefetch
# hand-over), interior and edge signal windows, the rolled general signal path (moving event), the four-tile inverse
B = 16384
La, Lir = int(4.4 * B), int(4.6 * B)
clips = [rng.standard_normal(La).astype(np.float32), rng.standard_normal(int(2.2 * B)).astype(np.float32)]
irs = (rng.standard_normal((2, 1 + 3, Lir)) * np.exp(-np.arange(Lir) / (Lir / 5.0))).astype(np.float32)
specs = [planning.EventSpec(n_samples=La, n_emitters=1, snr=10.0, emitter0=0),
         planning.EventSpec(n_samples=len(clips[1]), n_emitters=3, snr=12.0, emitter0=1, is_moving=True, duration=len(clips[1]) / 48000)]
pl = planning.plan_batch(specs, 2, Lir, 48000, log2_block=14)
for run_len in (0, 3):          # a batch this small gets runs of one partition; AL_FLAG_IR_RUN(3): 3 + 2, the hand-over between partitions
    setenv("AL_EXTRA_FLAGS", str(run_len << 24))
    batch = r.prepare(pl, clips, irs)
    assert batch.descs[0].flags & _hip.FLAG_QUAD_SPECTRA
    res = batch.run()
    res.check_finite()
    print("asan run ok: quad-tile transforms at B = 16384, runs of", run_len or 1, float(np.abs(res.spatial_audio(0)).sum()))
setenv("AL_EXTRA_FLAGS", None)
print("asan run ok: round-4 kernels")
