"""835 MB of pageable float32 (cfg2's IR tensor) -> HBM: Tensor.to, hipMemcpy, hipMemcpyAsync on a side stream (whole / in chunks),
from the main thread and from a helper thread while the main thread is busy in numpy (what Scene.generate does meanwhile)."""
import ctypes as ct, threading, time
import numpy as np, torch
rt = ct.CDLL("libamdhip64.so")
rt.hipMemcpyAsync.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int, ct.c_void_p]
rt.hipMemcpy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int]
rt.hipStreamSynchronize.argtypes = [ct.c_void_p]
n = 32 * 64 * 96000
src = np.random.default_rng(0).standard_normal(n).astype(np.float32)
dev = torch.empty(n, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
def t_to(): torch.from_numpy(src).to("cuda")
def t_sync(): rt.hipMemcpy(dev.data_ptr(), src.ctypes.data, src.nbytes, 1)
def t_async(chunks=1):
    step = (n // chunks + 3) // 4 * 4
    for a in range(0, n, step):
        b = min(n, a + step)
        rt.hipMemcpyAsync(dev.data_ptr() + 4 * a, src.ctypes.data + 4 * a, 4 * (b - a), 1, side.cuda_stream)
    rt.hipStreamSynchronize(side.cuda_stream)
def timed(fn, reps=6):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
def busy_main(fn):
    """fn on a helper thread while this thread copies 50 MB arrays around (the clip packing)."""
    a = np.zeros(12_000_000, np.float32); b = np.ones(12_000_000, np.float32)
    def run():
        th = threading.Thread(target=fn); th.start()
        while th.is_alive(): np.copyto(a, b)
        th.join()
    return run
for name, fn in [("Tensor.to", t_to), ("hipMemcpy", t_sync), ("hipMemcpyAsync side, 1 piece", t_async),
                 ("hipMemcpyAsync side, 4 pieces", lambda: t_async(4)), ("hipMemcpyAsync side, 16 pieces", lambda: t_async(16)),
                 ("hipMemcpyAsync side, 64 pieces", lambda: t_async(64))]:
    print(f"{name:34s} main thread {timed(fn):6.2f} ms   helper thread, main busy {timed(busy_main(fn)):6.2f} ms   "
          f"({src.nbytes / 1e9 / timed(fn) * 1e3:.1f} GB/s)", flush=True)
