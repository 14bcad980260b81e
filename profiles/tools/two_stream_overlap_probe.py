"""Round 4 probe: two resident cfg2 scenes on two HIP streams, the second one started half a scene later, so that the VALU-heavy
transform kernels of one overlap the HBM-bound accumulate / mixdown of the other -- against the same 2 x K steps on one stream."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from audiblelight_amd import synthetic

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
args = argparse.Namespace(log2_block=None, chunk_events=0, lanes=1)
r = bench.make_renderer(False) if hasattr(bench, "make_renderer") else None
if r is None:
    from audiblelight_amd import engine
    r = engine.Renderer()
scenes = [synthetic.make_scene(cfg, scene_index=i, torch_device="cuda") for i in range(2)]
res = [bench.resident_scene(r, sc, args) for sc in scenes]


def step(i):
    batch, mix = res[i][0], res[i][1]
    batch.run()
    mix.run()


K = 40
for i in (0, 1):
    step(i); step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    step(k % 2)
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / K * 1e3
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for offset in (0, 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if offset:                      # stream 1 starts with the cheap half of a step, so that the two run out of phase
        with torch.cuda.stream(streams[1]):
            res[1][0].run(stages=["al_forward_spectra", "al_emitter_gains", "al_spectral_mac"])
    for k in range(K):
        with torch.cuda.stream(streams[k % 2]):
            step(k % 2)
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / K * 1e3
    print(f"{cfg}: one stream {one:.3f} ms per step; two streams, {'out of phase' if offset else 'started together'}: {two:.3f} ms per step", flush=True)
