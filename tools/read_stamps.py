"""Development aid (GPU box, with an instrumented build as TSDR_HIP_LIB): per-workgroup phase durations of the raster kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tempest_loader import load_package
T = load_package()
import importlib
synth = importlib.import_module("tempestsdr_jl_amd.synth"); api = importlib.import_module("tempestsdr_jl_amd.api")
ctx = T.Context()
Fs, x_t, y_t, fv = 20e6, 2576, 1125, 60.0
S = synth.samples_per_frame(Fs, fv); nEch = 10_000_000; nb = nEch // S
iq = torch.from_numpy(synth.synth_leak(Fs, x_t, y_t, fv, nEch).view(np.float32)).cuda()
state = torch.zeros(480000, device="cuda"); fo = torch.empty(nb * 480000, device="cuda"); ro = torch.empty(nb * x_t * y_t, device="cuda")
si = torch.zeros(2 * nb, dtype=torch.int32, device="cuda"); sync = T.SyncXY(ctx, 600, 800); torch.cuda.synchronize()
for _ in range(5):
    api.frames_d(ctx, sync, iq, nEch, S, y_t, x_t, np.float32(0.1), True, state, fo, ro, si)
ctx.synchronize()
n = 5 * 8192
buf = np.zeros(n, np.uint64)
rc = ctx.lib.tsdr_debug_stamps(C.c_void_p(buf.ctypes.data), C.c_size_t(n))
st = buf.reshape(-1, 5).astype(np.int64)
st = st[st[:, 0] > 0]
d_stage_own = st[:, 1] - st[:, 0]; d_barrier = st[:, 2] - st[:, 1]; d_walk = st[:, 3] - st[:, 2]
print("workgroups stamped:", len(st), "(s_memtime ticks; 100 MHz realtime in col 4)")
for name, d in (("entry -> own staging done", d_stage_own), ("wait at the staging barrier", d_barrier), ("walk", d_walk), ("whole", st[:, 3] - st[:, 0])):
    print(f"{name:30s} mean {d.mean():9.0f}  p10 {np.percentile(d,10):9.0f}  median {np.median(d):9.0f}  p90 {np.percentile(d,90):9.0f}")
rt = st[:, 4]; print("kernel span by realtime (us):", (rt.max() - rt.min()) / 100.0, " ticks span:", st[:, 3].max() - st[:, 0].min())
