"""Fine-grained host timing of BatchDriver._stage at cfg2 (steady state: slots already page-locked)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from audiblelight_amd import batch as B, engine, synthetic, plan as planning

sc = synthetic.make_scene("cfg2")
r = engine.Renderer()
job = B.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs, starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name="s")
d = B.BatchDriver(r)
for _ in range(2):
    d._stage(job, 0)
torch.cuda.synchronize()
def T(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{label:44s} {1e3*best:7.2f} ms", flush=True); return out
pl = planning.plan_batch(job.specs, 32, sc.ir_len, sc.sr)
T("plan_batch+plan_mixdown", lambda: (planning.plan_batch(job.specs, 32, sc.ir_len, sc.sr), planning.plan_mixdown(job.starts, job.ends, [len(x) for x in job.clips], [32] * 64, pl.events["out_off"], list(range(64)), job.duration, job.sample_rate, 32)))
ah = d._pinned_buffer("audio", torch.float32, pl.audio_floats, 0)
T("pack_audio into pinned", lambda: r.pack_audio(pl, job.clips, out=ah.numpy()))
T("audio H2D", lambda: ah.to(r.mem.device, non_blocking=True))
T("upload_irs (current stream)", lambda: r.upload_irs(job.irs))
with torch.cuda.stream(d.copy_stream):
    T("upload_irs (copy stream)", lambda: r.upload_irs(job.irs))
flat = torch.from_numpy(job.irs.reshape(-1))
devbuf = torch.empty(flat.numel(), dtype=torch.float32, device="cuda")
T("devbuf.copy_(from_numpy)", lambda: devbuf.copy_(flat))
T("from_numpy.to(cuda)", lambda: flat.to("cuda"))
T("_stage whole", lambda: d._stage(job, 0))
st = d._stage(job, 0)
st = d._render(st)
T("_render (prepare+launch+mixdown)", lambda: d._render(st))
T("r.prepare only", lambda: r.prepare(st["plan"], job.clips, st["irs"], st["strides"], audio_dev=st["audio"]))
st2 = d._download(st, False, True, "PCM_16"); torch.cuda.synchronize()
T("_download scene (pinned reuse)", lambda: d._download(st, False, True, "PCM_16"))
T("_download frames pcm16", lambda: d._download(st, True, False, "PCM_16"))
T("_download frames pcm16 again", lambda: d._download(st, True, False, "PCM_16"))
for depth in (2, 4):
    dd = B.BatchDriver(r, depth=depth)
    dd.run([job] * 6, on_scene=lambda n, a: None, copy_for_callback=False)
    t0 = time.perf_counter(); dd.run([job] * 12, on_scene=lambda n, a: None, copy_for_callback=False); print(f"run x12 depth {depth}:", (time.perf_counter() - t0) / 12 * 1e3, "ms/scene")
