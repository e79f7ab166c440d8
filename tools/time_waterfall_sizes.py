"""Development aid (GPU box): getWaterfall of one C2 buffer (1e7 complex samples, inputs and outputs cycled) at several segment lengths."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tempest_loader import load_package
T = load_package()
ctx = T.Context()
p = lambda t: C.c_void_p(t.data_ptr())
L = 10_000_000
zs = [torch.randn(2 * L, dtype=torch.float32, device="cuda") for _ in range(3)]
wfs = [torch.empty(L, dtype=torch.float64, device="cuda") for _ in range(3)]
for n in (256, 512, 1024, 2048, 4096, 1000, 2000, 4000):
    nb = L // n
    def fn(i): ctx.call("tsdr_waterfall_d", p(zs[i % 3]), 1, L, n, p(wfs[i % 3]))
    for i in range(3): fn(i)
    ctx.synchronize(); t0 = time.perf_counter()
    for i in range(12): fn(i)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 12
    print(f"waterfall complex sizeFFT={n:5d} {dt*1e6:8.1f} us  {(8*L + 8*nb*n)/dt/1e9:7.1f} GB/s", flush=True)
