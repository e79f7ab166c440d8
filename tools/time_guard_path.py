"""Development aid (GPU box): what a flagged frame costs in the sync guard's launch -- per workload, per number of flagged
frames, x axis only (threshold 5e-3 on the box leak) and both axes (threshold 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tempest_loader import load_package
T = load_package()
import importlib
synth = importlib.import_module("tempestsdr_jl_amd.synth")
api = importlib.import_module("tempestsdr_jl_amd.api")
ctx = T.Context()
NPX = 600 * 800
for wl in (sys.argv[1:] or ["C2", "C3"]):
    w = synth.WORKLOADS[wl]
    Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
    S = synth.samples_per_frame(Fs, fv)
    nmax = 30
    iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nmax)
    d = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()
    state = torch.zeros(NPX, dtype=torch.float32, device="cuda")
    fo = torch.empty(nmax * NPX, dtype=torch.float32, device="cuda")
    ix = torch.zeros(nmax * 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for ppb, what in ((5_000_000, "x only"), (100_000_000, "both axes")):
        ctx.set_option("sync_guard_ppb", ppb)
        ctx.set_option("sync_guard_auto", 0)
        for nfr in (1, 2, 4, 8, 30):
            sync = T.SyncXY(ctx, 600, 800)
            for _ in range(3):
                api.frames_d(ctx, sync, d.data_ptr(), S * nfr, S, y_t, x_t, 0.1, True, state.data_ptr(), fo.data_ptr(), None, ix.data_ptr())
            ctx.synchronize()
            ctx.profile_reset(); ctx.profile(True)
            for _ in range(8):
                api.frames_d(ctx, sync, d.data_ptr(), S * nfr, S, y_t, x_t, 0.1, True, state.data_ptr(), fo.data_ptr(), None, ix.data_ptr())
            ctx.synchronize(); ctx.profile(False)
            p = ctx.profile_results()
            g = p["sync_guard"]["total_ms"] / p["sync_guard"]["launches"] * 1e3
            print(f"{wl} {what:9s} flagged frames {nfr:2d}: guard launch {g:7.1f} us = {g / nfr:6.1f} us per frame", flush=True)
    ctx.set_option("sync_guard_ppb", 20000); ctx.set_option("sync_guard_auto", 1)
