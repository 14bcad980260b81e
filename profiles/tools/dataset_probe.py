"""render_dataset on many SMALL scenes (examples/render_dataset.py: 10 s, 4 capsules, 2-5 events): where a scene's host time goes."""
import cProfile, pstats, sys, os, time, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "examples"))
import render_dataset as ex
from audiblelight_amd import batch
out = "/tmp/ds_prof"
shutil.rmtree(out, ignore_errors=True)
drv = batch.BatchDriver()
batch.render_dataset(((f"w{i}", (lambda i=i: ex.make_scene(i))) for i in range(12)), out, driver=drv)   # warm
t0 = time.perf_counter()
rep = batch.render_dataset(((f"s{i}", (lambda i=i: ex.make_scene(i))) for i in range(60)), out, driver=drv)
dt = time.perf_counter() - t0
print("60 scenes", round(dt, 2), "s", round(dt / 60 * 1e3, 1), "ms/scene", {k: round(v / 60 * 1e3, 2) for k, v in rep.host_s.items()})
# the scene source alone, in this thread
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(20):
    sc = ex.make_scene(100 + i)
t1 = time.perf_counter()
jobs = []
for i in range(20):
    sc = ex.make_scene(100 + i)
    jobs.append(batch.scene_jobs(sc, f"x{i}", drv.r))
t2 = time.perf_counter()
pr.disable()
print("make_scene %.1f ms, scene_jobs %.1f ms per scene" % ((t1 - t0) / 20 * 1e3, ((t2 - t1) - (t1 - t0)) / 20 * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
# the synchronous API in a loop (what a reference user's script does): Scene.generate per scene
for i in range(4):
    ex.make_scene(200 + i).generate(output_dir=os.path.join(out, f"g{i}"), metadata_dcase=False)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(30):
    ex.make_scene(300 + i).generate(output_dir=os.path.join(out, f"h{i}"), metadata_dcase=False)
dt = time.perf_counter() - t0
pr.disable()
print("Scene.generate loop: %.1f ms/scene" % (dt / 30 * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
