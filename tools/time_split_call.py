"""Development aid (GPU box): one tsdr_frames_d per buffer against the same buffer cut into CH chunks of frames that go through
the submit pipeline and are flushed before the step ends (no overlap ACROSS steps): what an intra-call split would give.
   python tools/time_split_call.py [steps] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tempest_loader import load_package
T = load_package()
import importlib
synth = importlib.import_module("tempestsdr_jl_amd.synth")
api = importlib.import_module("tempestsdr_jl_amd.api")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
wl = sys.argv[2] if len(sys.argv) > 2 else "C2"
w = synth.WORKLOADS[wl]
Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
S = synth.samples_per_frame(Fs, fv); nfr = int(round(w["acquisition"] * Fs)) // S; NPX = 600 * 800; P = x_t * y_t
iqs = [torch.from_numpy(np.ascontiguousarray(synth.synth_leak(Fs, x_t, y_t, fv, S * nfr, n0=b * S * nfr)).view(np.float32)).cuda() for b in range(3)]
ctx = T.Context()
for raster in (True, False):
    for ch in (1, 2, 3):
        sync = T.SyncXY(ctx, 600, 800)
        state = torch.zeros(NPX, dtype=torch.float32, device="cuda")
        outs = [(torch.empty(nfr * NPX, dtype=torch.float32, device="cuda"),
                 torch.empty(nfr * P, dtype=torch.float32, device="cuda") if raster else None,
                 torch.zeros(nfr * 2, dtype=torch.int32, device="cuda")) for _ in range(2)]
        per = nfr // ch
        assert per * ch == nfr

        def run(n):
            for i in range(n):
                fo, ra, ix = outs[i & 1]
                iq = iqs[i % 3]
                if ch == 1:
                    api.frames_d(ctx, sync, iq.data_ptr(), S * nfr, S, y_t, x_t, 0.1, True, state.data_ptr(), fo.data_ptr(),
                                 ra.data_ptr() if ra is not None else None, ix.data_ptr())
                else:
                    for c in range(ch):
                        api.frames_submit_d(ctx, sync, iq.data_ptr() + 8 * c * per * S, per * S, S, y_t, x_t, 0.1, True, state.data_ptr(),
                                            fo.data_ptr() + 4 * c * per * NPX, ra.data_ptr() + 4 * c * per * P if ra is not None else None,
                                            ix.data_ptr() + 8 * c * per)
                    api.frames_flush(ctx)
                    # what the next call's entry does: its lanes wait for the context's stream
                    ctx.call("tsdr_abs2_d", iq.data_ptr(), 64, fo.data_ptr()) if False else None
            ctx.synchronize()
        run(20)
        res = []
        for _ in range(3):
            t0 = time.perf_counter()
            run(steps)
            res.append((time.perf_counter() - t0) / steps * 1e3)
        res.sort()
        print(f"{wl} raster={raster} chunks={ch}: {res[1]:.4f} ms per buffer = {nfr / res[1]:8.1f} k frames/s", flush=True)
        sync.close()
