"""Development aid (GPU box): the frame loop's raster launch with pieces of its work switched off through the API --
do_align off (no projection sums in the walk), raster off (no column stores) -- to see what each piece costs."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tempest_loader import load_package
T = load_package()
synth = importlib.import_module("tempestsdr_jl_amd.synth")
api = importlib.import_module("tempestsdr_jl_amd.api")
ctx = T.Context()
w = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
S = synth.samples_per_frame(Fs, fv); nEch = int(round(0.5 * Fs)); nb = nEch // S
iq = [torch.from_numpy(synth.synth_leak(Fs, x_t, y_t, fv, nEch, n0=b * nEch).view(np.float32)).cuda() for b in range(3)]
state = torch.zeros(480000, device="cuda"); fo = torch.empty(nb * 480000, device="cuda")
ro = torch.empty(nb * x_t * y_t, device="cuda"); si = torch.zeros(2 * nb, dtype=torch.int32, device="cuda")
sync = T.SyncXY(ctx, 600, 800)
torch.cuda.synchronize()
for align in (True, False):
    for raster in (True, False):
        def step(i):
            api.frames_d(ctx, sync, iq[i % 3], nEch, S, y_t, x_t, np.float32(0.1), align, state, fo, ro if raster else None, si if align else None)
        for i in range(5): step(i)
        ctx.synchronize()
        ctx.profile_reset(); ctx.profile(True)
        for i in range(30): step(i)
        ctx.synchronize(); ctx.profile(False)
        pr = ctx.profile_results()
        print(f"align={align} raster={raster}:", {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in sorted(pr.items())})
