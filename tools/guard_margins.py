import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tempest_loader import load_package
T = load_package()
synth = importlib.import_module("tempestsdr_jl_amd.synth"); api = importlib.import_module("tempestsdr_jl_amd.api")
ctx = T.Context()
for wl in ("C3", "C5"):
    w = synth.WORKLOADS[wl]; Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
    S = synth.samples_per_frame(Fs, fv); nEch = int(round(0.5 * Fs)); nb = nEch // S
    iq = torch.from_numpy(synth.synth_leak(Fs, x_t, y_t, fv, nEch).view(np.float32)).cuda()
    state = torch.zeros(480000, device="cuda"); fo = torch.empty(nb * 480000, device="cuda"); si = torch.zeros(2 * nb, dtype=torch.int32, device="cuda")
    sync = T.SyncXY(ctx, 600, 800); torch.cuda.synchronize()
    ctx.sync_guard_stats(reset=True)
    api.frames_d(ctx, sync, iq, nEch, S, y_t, x_t, np.float32(0.1), True, state, fo, None, si)
    ctx.synchronize()
    m = ctx.sync_guard_margins()
    print(wl, ctx.sync_guard_stats(), "margins x:", np.array2string(m[:, 0], precision=2), "y min", m[:, 1].min())
    print(si.cpu().numpy().reshape(-1, 2)[:, 1])
