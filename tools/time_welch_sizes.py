"""Development aid (GPU box): getWelch of one C2 buffer (1e7 complex samples, inputs cycled) at several segment lengths."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tempest_loader import load_package
T = load_package()
ctx = T.Context()
p = lambda t: C.c_void_p(t.data_ptr())
L = 10_000_000
zs = [torch.randn(2 * L, dtype=torch.float32, device="cuda") for _ in range(4)]   # 320 MB cycled: cold MALL
def timeit(name, fn, reps=12):
    for i in range(3): fn(i)
    ctx.synchronize(); t0 = time.perf_counter()
    for i in range(reps): fn(i)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{name:30s} {dt*1e6:8.1f} us  {8*L/dt/1e9:7.1f} GB/s", flush=True)
for n in (128, 256, 512, 1024, 2048, 4096, 500, 1000, 2000, 4000, 768, 960, 1200, 1280, 1600, 2500, 3200, 3000):
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    timeit(f"welch complex sizeFFT={n}", lambda i: ctx.call("tsdr_welch_d", p(zs[i % 4]), 1, L, n, 0, p(y)))
