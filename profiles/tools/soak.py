"""Soak: many scenes through the pipelined driver; device / host memory must stay flat and nothing may hang."""
import os, sys, time, resource
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "examples"))
import numpy as np, torch, shutil
import render_dataset as ex
from audiblelight_amd import batch as B, engine, synthetic
out = "/tmp/soak_out"; shutil.rmtree(out, ignore_errors=True)
if os.environ.get("SOAK_NO_GC") == "1":     # device buffers must come back by reference count alone
    import gc; gc.disable(); print("garbage collector disabled", flush=True)
drv = B.BatchDriver()
def mem(): return round(torch.cuda.memory_reserved() / 1e9, 2), round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2)
for rnd in range(4):
    t0 = time.perf_counter()
    rep = B.render_dataset(((f"r{rnd}_s{i:03d}", (lambda i=i: ex.make_scene(1000 * rnd + i))) for i in range(150)), out, driver=drv)
    print(f"round {rnd}: 150 small scenes in {time.perf_counter() - t0:.2f} s; reserved GB / max RSS GB", mem(), flush=True)
    shutil.rmtree(out, ignore_errors=True)
sc = synthetic.make_scene("cfg2")
jobs = lambda n: (B.SceneJob(specs=sc.specs, clips=sc.clips, irs=sc.irs if i % 2 else sc.irs.astype(np.float64), starts=sc.starts, ends=sc.ends,
                             duration=sc.duration, sample_rate=sc.sr, name=f"big{i}") for i in range(n))
for rnd in range(3):
    t0 = time.perf_counter()
    rep = drv.run(jobs(24), output_dir=out)
    print(f"round {rnd}: 24 cfg2 scenes (float32 / float64 IRs alternating) to PCM_16 files in {time.perf_counter() - t0:.2f} s; reserved GB / max RSS GB", mem(), flush=True)
    shutil.rmtree(out, ignore_errors=True)
print("soak ok")
