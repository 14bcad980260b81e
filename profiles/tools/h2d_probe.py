"""How fast can 786 MB of caller-owned float32 IRs reach HBM?  pageable .to(), pinned staging (1 / N copy threads),
hipHostRegister on the caller's buffer (registration cost + async copy)."""
import ctypes as ct
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

N = 32 * 64 * 96000
src = np.random.default_rng(0).standard_normal(N, dtype=np.float32)
dev = torch.empty(N, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()


def timed(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{label:60s} {best * 1e3:8.1f} ms  {N * 4 / best / 1e9:6.1f} GB/s", flush=True)


t = torch.from_numpy(src)
timed("pageable tensor.to(cuda) via dev.copy_", lambda: dev.copy_(t))
pinned = torch.empty(N, dtype=torch.float32).pin_memory()
timed("host memcpy into pinned staging, 1 thread", lambda: pinned.copy_(t))
for workers in (4, 8, 16, 32):
    pool = ThreadPoolExecutor(workers)
    step = -(-N // workers)

    def par():
        futs = [pool.submit(lambda a, b: pinned[a:b].copy_(t[a:b]), i, min(i + step, N)) for i in range(0, N, step)]
        [f.result() for f in futs]
    timed(f"host memcpy into pinned staging, {workers} threads", par)
timed("pinned -> device (async DMA)", lambda: dev.copy_(pinned, non_blocking=True))
rt = torch.cuda.cudart()
t0 = time.perf_counter()
rc = rt.cudaHostRegister(src.ctypes.data, N * 4, 0)
t_reg = time.perf_counter() - t0
print(f"hipHostRegister rc={rc} {t_reg * 1e3:.1f} ms", flush=True)
if int(rc) == 0:
    timed("registered caller buffer -> device", lambda: dev.copy_(t, non_blocking=True))
    t0 = time.perf_counter()
    rt.cudaHostUnregister(src.ctypes.data)
    print(f"hipHostUnregister {(time.perf_counter() - t0) * 1e3:.1f} ms")
    src2 = np.random.default_rng(1).standard_normal(N, dtype=np.float32)
    t0 = time.perf_counter()
    rt.cudaHostRegister(src2.ctypes.data, N * 4, 0)
    print(f"hipHostRegister of a second (touched) buffer {(time.perf_counter() - t0) * 1e3:.1f} ms")
# D2H of a scene
out_dev = torch.empty(32 * 2880000, dtype=torch.float32, device="cuda")
out_pin = torch.empty(32 * 2880000, dtype=torch.float32).pin_memory()
M = out_dev.numel()
torch.cuda.synchronize(); t0 = time.perf_counter(); out_pin.copy_(out_dev, non_blocking=True); torch.cuda.synchronize()
print(f"D2H scene into pinned: {(time.perf_counter() - t0) * 1e3:.1f} ms {M * 4 / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
torch.cuda.synchronize(); t0 = time.perf_counter(); h = out_dev.cpu(); torch.cuda.synchronize()
print(f"D2H scene .cpu() pageable: {(time.perf_counter() - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); z = np.zeros(N, dtype=np.float32); z[:] = src; print(f"np.zeros + copy {(time.perf_counter() - t0) * 1e3:.1f} ms")
import os; print("cpus", os.cpu_count())
