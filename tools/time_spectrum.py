"""Development aid (GPU box): device-resident timing of getSpectrum / getWelch / getWaterfall / resampler! at the C2 sizes
(algorithmic bytes: 8*L in + output)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tempest_loader import load_package
T = load_package()
ctx = T.Context()
p = lambda t: C.c_void_p(t.data_ptr())


def timeit(name, fn, nbytes, reps=20):
    for _ in range(3): fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:34s} {dt*1e6:9.1f} us  {nbytes/dt/1e9:8.1f} GB/s of algorithmic bytes")
    return dt


L = 10_000_000
z = torch.randn(2 * L, dtype=torch.float32, device="cuda")
y = torch.empty(1024, dtype=torch.float32, device="cuda")
nb = L // 1024
wf = torch.empty(nb * 1024, dtype=torch.float64, device="cuda")
timeit("welch 1e7 complex, 1024", lambda: ctx.call("tsdr_welch_d", p(z), 1, L, 1024, 0, p(y)), 8 * L + 4096)
timeit("welch 1e7 real, 1024", lambda: ctx.call("tsdr_welch_d", p(z), 0, L, 1024, 0, p(y)), 4 * L + 4096)
timeit("waterfall 1e7 complex, 1024", lambda: ctx.call("tsdr_waterfall_d", p(z), 1, L, 1024, p(wf)), 8 * L + 8 * nb * 1024)
timeit("welch 1e7 complex, 1000 (generic)", lambda: ctx.call("tsdr_welch_d", p(z), 1, L, 1000, 0, p(y)), 8 * L + 4000)
ys = torch.empty(80000, dtype=torch.float32, device="cuda")
timeit("spectrum N=80000 complex", lambda: ctx.call("tsdr_spectrum_d", p(z), 1, 80000, 0, p(ys)), 8 * 80000 + 4 * 80000, reps=50)
