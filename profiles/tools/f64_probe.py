"""float64 IR tensors (what the reference's WorldState.get_irs() returns): ways to get them into HBM as float32."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
n = 32 * 64 * 96000
src = np.random.default_rng(0).standard_normal(n)            # float64, pageable
dev = torch.empty(n, dtype=torch.float32, device="cuda")
dev64 = torch.empty(n, dtype=torch.float64, device="cuda")
pin = torch.empty(n, dtype=torch.float32).pin_memory()
def T(label, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f"{label:64s} {(time.perf_counter() - t0) / reps * 1e3:7.1f} ms", flush=True)
T("upload float64 as it is (pageable, blocking) + device convert", lambda: dev.copy_(dev64.copy_(torch.from_numpy(src))))
T("numpy astype(float32), one thread", lambda: src.astype(np.float32), reps=1)
for nt in (4, 8, 16, 32, 64):
    pool = ThreadPoolExecutor(nt)
    bounds = np.linspace(0, n, nt + 1).astype(np.int64)
    view = pin.numpy()
    def conv():
        list(pool.map(lambda i: np.copyto(view[bounds[i]: bounds[i + 1]], src[bounds[i]: bounds[i + 1]], casting="same_kind"), range(nt)))
    T(f"convert into page-locked float32, {nt} threads", conv)
    def conv_up(chunks=8):
        cb = np.linspace(0, n, chunks + 1).astype(np.int64)
        for c in range(chunks):
            sub = np.linspace(cb[c], cb[c + 1], nt + 1).astype(np.int64)
            list(pool.map(lambda i: np.copyto(view[sub[i]: sub[i + 1]], src[sub[i]: sub[i + 1]], casting="same_kind"), range(nt)))
            dev[cb[c]: cb[c + 1]].copy_(pin[cb[c]: cb[c + 1]], non_blocking=True)
    T(f"  ... in 8 chunks, each DMA'd while the next converts, {nt} threads", conv_up)
    pool.shutdown()
