"""Per-kernel HIP-event breakdown of the configuration search (development aid; GPU box only)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tempest_loader import load_package
T = load_package()

ctx = T.Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
k_hi = n // 2
Fs = 200e6
g = torch.Generator().manual_seed(1)
iq = torch.view_as_real(torch.randn(n, dtype=torch.complex64, generator=g)).contiguous().cuda()
out = torch.empty(k_hi, dtype=torch.float32, device="cuda")
n_out = C.c_size_t(0)
mode = sys.argv[2] if len(sys.argv) > 2 else "search"   # "search": tsdr_autocorr_search_d (lags + fused findmax); "lags": lags only
if len(sys.argv) > 3:
    ctx.set_option("ac_fuse_mid", int(sys.argv[3]))
lo, cnt = k_hi // 9, k_hi - k_hi // 9
idx, val = C.c_size_t(0), C.c_float(0)
def once():
    if mode == "lags":
        ctx.call("tsdr_autocorr_iq_d", C.c_void_p(iq.data_ptr()), n, Fs, 0.0, k_hi / Fs, 1, C.c_void_p(out.data_ptr()), C.byref(n_out))
    else:
        ctx.call("tsdr_autocorr_search_d", C.c_void_p(iq.data_ptr()), 1, n, Fs, 0.0, k_hi / Fs, 1, C.c_void_p(out.data_ptr()), C.byref(n_out),
                 lo, cnt, C.byref(idx), C.byref(val))
for _ in range(3): once()
ctx.synchronize()
ctx.profile(True); ctx.profile_reset()
for _ in range(10): once()
ctx.synchronize()
tot = 0
for k, v in sorted(ctx.profile_results().items()):
    cnt, ms = v["launches"], v["total_ms"]
    print(f"{k:16s} n={cnt:4d} avg={ms/cnt*1e3:8.2f} us  per-search={ms/10*1e3:8.2f} us"); tot += ms / 10
print("total per search", round(tot * 1e3, 2), "us")
# back-to-back searches without per-kernel events or host readback: GPU-side time per search incl. launch gaps
import time
ctx.profile(False)
for _ in range(3): once()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(50): once()
ctx.synchronize()
print("back-to-back (search mode: with the findmax readback):", round((time.perf_counter() - t0) / 50 * 1e6, 1), "us per search")
