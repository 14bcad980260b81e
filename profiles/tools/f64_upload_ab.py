"""Renderer.upload_irs on cfg2's IR tensor as float64: host thread-pool cast + chunked DMA (default) vs float64 upload + device cast."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiblelight_amd import switches as _sw   # AL_* switches are parsed once per process: set them through set_env
import numpy as np, torch
from audiblelight_amd import engine, synthetic, batch as B
sc = synthetic.make_scene("cfg2")
irs64 = sc.irs.astype(np.float64)
r = engine.Renderer()
for mode in ("device", "host", "device", "host"):
    _sw.set_env("AL_F64_UPLOAD", mode)
    for _ in range(2): d, _s = r.upload_irs(irs64); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): d, _s = r.upload_irs(irs64)
    torch.cuda.synchronize()
    print(f"upload_irs float64 ({mode} cast): {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms", flush=True)
ref, _ = r.upload_irs(sc.irs); _sw.set_env("AL_F64_UPLOAD", "host"); got, _ = r.upload_irs(irs64)
print("same bits as the float32 upload:", bool(torch.equal(ref, got)))
jobs = [B.SceneJob(specs=sc.specs, clips=sc.clips, irs=irs64, starts=sc.starts, ends=sc.ends, duration=sc.duration, sample_rate=sc.sr, name=f"s{i}") for i in range(12)]
for mode in ("device", "host"):
    _sw.set_env("AL_F64_UPLOAD", mode)
    drv = B.BatchDriver(r)
    drv.run(jobs[:6], on_scene=lambda n, a: None, copy_for_callback=False)
    t0 = time.perf_counter(); rep = drv.run(jobs, on_scene=lambda n, a: None, copy_for_callback=False); dt = time.perf_counter() - t0
    print(f"batch driver, float64 IRs ({mode} cast): {dt / 12 * 1e3:.1f} ms/scene", {k: round(v / 12 * 1e3, 1) for k, v in rep.host_s.items()}, flush=True)
