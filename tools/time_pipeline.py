"""Development aid (GPU box): the frame loop one call per buffer (tsdr_frames_d) against the two-lane pipeline
(tsdr_frames_submit_d) on C2, raster and raster-free: ms per buffer, and how much of it the HOST needs to enqueue a buffer
(the loop's return time without a synchronisation).   python tools/time_pipeline.py [steps] [workload]"""
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
# DUMMY_STREAMS=K: K HIP streams created (and kept) before the context and its lanes exist -- does the pipeline's overlap
# depend on which hardware queues its lanes are mapped to?
_dummy = []
if int(os.environ.get("DUMMY_STREAMS", "0")):
    import ctypes
    _hip = ctypes.CDLL("libamdhip64.so.7")
    for _ in range(int(os.environ["DUMMY_STREAMS"])):
        st = ctypes.c_void_p()
        assert _hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        _dummy.append(st)
ctx = T.Context()
for raster in (True, False):
    for pipe in (False, True):
        sync = T.SyncXY(ctx, 600, 800)
        state = torch.zeros(NPX, dtype=torch.float32, device="cuda")
        outs = [(torch.empty(nfr * NPX, dtype=torch.float32, device="cuda"),
                 torch.empty(nfr * P, dtype=torch.float32, device="cuda") if raster else None,
                 torch.zeros(nfr * 2, dtype=torch.int32, device="cuda")) for _ in range(2)]
        f = api.frames_submit_d if pipe else api.frames_d

        def run(n):
            for i in range(n):
                fo, ra, ix = outs[i & 1]
                f(ctx, sync, iqs[i % 3].data_ptr(), S * nfr, S, y_t, x_t, 0.1, True, state.data_ptr(), fo.data_ptr(),
                  ra.data_ptr() if ra is not None else None, ix.data_ptr())
            t = time.perf_counter()
            if pipe:
                api.frames_flush(ctx)
            ctx.synchronize()
            return t
        run(20)
        if pipe:   # the library times its candidate arrangements on the first submissions of a configuration
            while ctx.pipeline_info()["trials_left"] > 0:
                run(10)
            print("   ", ctx.pipeline_info()["text"], flush=True)
        res = []
        for _ in range(3):
            t0 = time.perf_counter()
            th = run(steps)
            t1 = time.perf_counter()
            res.append(((t1 - t0) / steps * 1e3, (th - t0) / steps * 1e3))
        res.sort()
        ms, host = res[1]
        print(f"{wl} raster={raster} pipeline={pipe}: {ms:.4f} ms per buffer = {nfr / ms:8.1f} k frames/s; host enqueue {host:.4f} ms per buffer", flush=True)
        sync.close()
