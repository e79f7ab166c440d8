"""Development aid (GPU box): device-resident timing of tsdr_fft_c2c_d at given lengths (default: the search transforms
of C2 / C5 / C3).   python tools/time_fft.py [N ...]   -- run under rocprofv3 --kernel-trace --stats for per-pass times."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tempest_loader import load_package
T = load_package()
ctx = T.Context()
p = lambda t: C.c_void_p(t.data_ptr())
sizes = [int(float(a)) for a in sys.argv[1:]] or [2_000_000, 5_000_000, 20_000_000, 1 << 21, 1 << 24]
for n in sizes:
    x = torch.randn(2 * n, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    reps = 20
    for _ in range(3): ctx.call("tsdr_fft_c2c_d", p(x), p(y), n, 1, -1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): ctx.call("tsdr_fft_c2c_d", p(x), p(y), n, 1, -1)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"fft n={n:>9d}  {dt*1e6:9.1f} us   {16*n/dt/1e9:8.1f} GB/s per pass-equivalent (16 B/point)")
