"""Development aid (GPU box): achieved GB/s of the elementwise / FFT entry points on device-resident data."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tempest_loader import load_package
T = load_package()
ctx = T.Context()
n = 100_000_000
iq = torch.randn(2 * n, dtype=torch.float32, device="cuda")
out = torch.empty(n, dtype=torch.float32, device="cuda")

def timeit(name, fn, nbytes, reps=20):
    for _ in range(3): fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:28s} {dt*1e6:9.1f} us  {nbytes/dt/1e9:8.1f} GB/s")

p = lambda t: C.c_void_p(t.data_ptr())
timeit("am_demod 1e8", lambda: ctx.call("tsdr_am_demod_d", p(iq), n, p(out)), 12 * n)
timeit("abs2 1e8", lambda: ctx.call("tsdr_abs2_d", p(iq), n, p(out)), 12 * n)
timeit("invert_am 1e8", lambda: ctx.call("tsdr_invert_am_d", p(iq), n, p(out)), 12 * n)
timeit("fm_demod 1e8", lambda: ctx.call("tsdr_fm_demod_d", p(iq), n, p(out)), 12 * n)
img = torch.rand(1125 * 2576, dtype=torch.float32, device="cuda")
dn = torch.empty(600 * 800, dtype=torch.float32, device="cuda")
timeit("downgrade 1125x2576", lambda: ctx.call("tsdr_resize2d_d", p(img), 1125, 2576, 600, 800, p(dn)), 4 * (1125 * 2576 + 480000), reps=200)
for N in (1 << 22, 4_000_000, 1 << 24):
    z = torch.randn(2 * N, dtype=torch.float32, device="cuda"); zo = torch.empty_like(z)
    timeit(f"fft_c2c N={N}", lambda: ctx.call("tsdr_fft_c2c_d", p(z), p(zo), N, 1, -1), 16 * N, reps=50)
zb = torch.randn(2 * 1024 * 4096, dtype=torch.float32, device="cuda"); zbo = torch.empty_like(zb)
timeit("fft_c2c 1024 x 4096 rows", lambda: ctx.call("tsdr_fft_c2c_d", p(zb), p(zbo), 1024, 4096, -1), 16 * 1024 * 4096, reps=50)
