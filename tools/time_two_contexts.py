"""Development aid (GPU box): aggregate frames/s of N contexts (one HIP stream each, one host thread each) running the frame loop
on their own resident buffers side by side on ONE GPU -- does a second capture stream fill the first one's gaps?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tempest_loader import load_package
T = load_package()
import importlib
synth = importlib.import_module("tempestsdr_jl_amd.synth")
api = importlib.import_module("tempestsdr_jl_amd.api")
w = synth.WORKLOADS["C2"]
Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
S = synth.samples_per_frame(Fs, fv); nfr = 30; NPX = 600 * 800; P = x_t * y_t
iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr)
steps = 300


class Worker:
    def __init__(self, raster):
        self.ctx = T.Context()
        self.sync = T.SyncXY(self.ctx, 600, 800)
        self.d = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()
        self.state = torch.zeros(NPX, dtype=torch.float32, device="cuda")
        self.fo = torch.empty(nfr * NPX, dtype=torch.float32, device="cuda")
        self.ra = torch.empty(nfr * P, dtype=torch.float32, device="cuda") if raster else None
        self.ix = torch.zeros(nfr * 2, dtype=torch.int32, device="cuda")

    def step(self):
        api.frames_d(self.ctx, self.sync, self.d.data_ptr(), S * nfr, S, y_t, x_t, 0.1, True, self.state.data_ptr(), self.fo.data_ptr(),
                     self.ra.data_ptr() if self.ra is not None else None, self.ix.data_ptr())

    def run(self, n):
        for _ in range(n):
            self.step()
        self.ctx.synchronize()


for raster in (True, False):
    for nctx in (1, 2, 3):
        ws = [Worker(raster) for _ in range(nctx)]
        torch.cuda.synchronize()
        for x in ws:
            x.run(20)
        t0 = time.perf_counter()
        th = [threading.Thread(target=x.run, args=(steps,)) for x in ws]
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        print(f"raster={raster} contexts={nctx}: {nctx * steps * nfr / dt:10.0f} frames/s aggregate ({dt / steps * 1e3:.4f} ms per round of {nctx} buffers)", flush=True)
        del ws
        torch.cuda.empty_cache()
