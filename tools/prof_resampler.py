import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tempest_loader import load_package
T = load_package(); ctx = T.Context(); lib = T._lib.load()
nb, up = 1_000_000, 4
r = C.c_void_p(0); assert lib.tsdr_resampler_init(ctx.h, nb, up, C.byref(r)) == 0
x = [torch.randn(nb, device="cuda") for _ in range(16)]
o = [torch.empty(nb * up, device="cuda") for _ in range(16)]
torch.cuda.synchronize()
def once(i): assert lib.tsdr_resampler_run_d(r, C.c_void_p(x[i % 16].data_ptr()), nb, C.c_void_p(o[i % 16].data_ptr())) == 0
for i in range(4): once(i)
ctx.synchronize(); ctx.profile_reset(); ctx.profile(True)
for i in range(32): once(i)
ctx.synchronize(); ctx.profile(False)
tot = 0
for k, v in ctx.profile_results().items():
    print(f"{k:20s} n={v['launches']:4d} avg {v['total_ms']/v['launches']*1e3:7.2f} us"); tot += v['total_ms'] / 32 * 1e3
print("sum per call", round(tot, 1), "us")
