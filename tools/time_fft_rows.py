"""Development aid (GPU box): batched row transforms (tsdr_fft_c2c_d) of 1e7 points at several row lengths."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tempest_loader import load_package
T = load_package()
ctx = T.Context()
p = lambda t: C.c_void_p(t.data_ptr())
L = 10_000_000
z = torch.randn(2 * L, dtype=torch.float32, device="cuda"); zo = torch.empty_like(z)
for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 1000, 2000, 500, 250, 100):
    b = L // n
    fn = lambda: ctx.call("tsdr_fft_c2c_d", p(z), p(zo), n, b, -1)
    for _ in range(3): fn()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"fft rows n={n:5d} x {b:7d}: {dt*1e6:8.1f} us  {16*n*b/dt/1e9:7.1f} GB/s", flush=True)
