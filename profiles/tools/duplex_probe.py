"""Do H2D of a caller buffer (786 MB) and D2H of a scene (369 MB) overlap?  pageable blocking vs registered async."""
import time
import numpy as np, torch

N = 32 * 64 * 96000; M = 32 * 2880000
src = np.random.default_rng(0).standard_normal(N, dtype=np.float32); t = torch.from_numpy(src)
dev = torch.empty(N, dtype=torch.float32, device="cuda")
out_dev = torch.randn(M, device="cuda"); out_pin = torch.empty(M, dtype=torch.float32).pin_memory()
s_up, s_down = torch.cuda.Stream(), torch.cuda.Stream()
def run(label, up, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(s_down):
            out_pin.copy_(out_dev, non_blocking=True)
        with torch.cuda.stream(s_up):
            up()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{label:52s} {best*1e3:7.2f} ms", flush=True)
run("D2H(pinned, async) + H2D pageable blocking", lambda: dev.copy_(t))
rt = torch.cuda.cudart(); t0 = time.perf_counter(); rc = rt.cudaHostRegister(src.ctypes.data, N * 4, 0); print("register ms", (time.perf_counter() - t0) * 1e3, rc)
run("D2H(pinned, async) + H2D registered async", lambda: dev.copy_(t, non_blocking=True))
def only_up():
    torch.cuda.synchronize(); t0 = time.perf_counter(); dev.copy_(t, non_blocking=True); torch.cuda.synchronize(); return time.perf_counter() - t0
print("H2D registered alone", min(only_up() for _ in range(3)) * 1e3)
# with a compute kernel running too
x = torch.randn(1 << 28, device="cuda")
def up_with_compute():
    torch.cuda.current_stream().wait_stream(s_up)
    for _ in range(6): x.mul_(1.0001)
    dev.copy_(t, non_blocking=True)
run("D2H + H2D registered + streaming kernels", up_with_compute)
t0 = time.perf_counter(); rt.cudaHostUnregister(src.ctypes.data); print("unregister ms", (time.perf_counter() - t0) * 1e3)
# registration cost for repeated register/unregister of the same buffer
for _ in range(3):
    t0 = time.perf_counter(); rt.cudaHostRegister(src.ctypes.data, N * 4, 0); a = time.perf_counter() - t0
    t0 = time.perf_counter(); rt.cudaHostUnregister(src.ctypes.data); b = time.perf_counter() - t0
    print(f"re-register {a*1e3:.2f} ms unregister {b*1e3:.2f} ms")
