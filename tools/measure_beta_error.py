"""Development aid (GPU box): how far FAST-mode beta is from EXACT-mode beta (= the oracle's, bit for bit).

Every frame of a few capture buffers goes through the frame loop alone (one-frame calls, so that the beta matrices of
that frame can be read back) in TSDR_FAST with the sync guard off and in TSDR_EXACT; printed: the largest relative
difference of the per-column maxima (the quantity the frame-sync decision and the guard's margin are made of) and of
the whole matrices.  The guard threshold (2e-5) has to stay well above twice the former.

    python tools/measure_beta_error.py [C2|C3|C5|T] [frames] [raster|noraster]   (T: the 1056x628 @ 2 MS/s test geometry;
    the raster-writing and the raster-free FAST paths are different kernels: measure both)
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tempest_loader import load_package

T = load_package()
synth = importlib.import_module("tempestsdr_jl_amd.synth")


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    want_raster = len(sys.argv) > 3 and sys.argv[3] == "raster"
    if wl == "T":
        Fs, x_t, y_t, fv = 2.0e6, 1056, 628, 60.0
    else:
        w = synth.WORKLOADS[wl]
        Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
    S = synth.samples_per_frame(Fs, fv)
    ctx = T.Context()
    worst_cm, worst_all = 0.0, 0.0
    for card in ("box", "plateau"):
        iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr, card=card)
        for f in range(nfr):
            z = iq[f * S:(f + 1) * S]
            b = {}
            for mode in ("fast", "exact"):
                ctx.set_precision(mode)
                ctx.set_option("sync_guard_ppb", 0)
                try:
                    sync = T.SyncXY(ctx, 600, 800)
                    st = np.zeros((600, 800), np.float32, order="F")
                    ctx.frames(sync, z, S, y_t, x_t, np.float32(0.1), st, want_frames=False, want_raster=want_raster)
                    b[mode] = (sync.beta("x").astype(np.float64), sync.beta("y").astype(np.float64))
                    sync.close()
                finally:
                    ctx.set_precision("fast")
                    ctx.set_option("sync_guard_ppb", 20000)
            for a, e in zip(b["fast"], b["exact"]):
                cm_a, cm_e = a.max(axis=0), e.max(axis=0)
                worst_cm = max(worst_cm, float(np.max(np.abs(cm_a - cm_e) / cm_e)))
                worst_all = max(worst_all, float(np.max(np.abs(a - e) / e)))
        print(f"{wl} {'raster' if want_raster else 'raster-free'} {card}: after {nfr} frames: max rel diff of column maxima {worst_cm:.3e}, of all beta values {worst_all:.3e}")


if __name__ == "__main__":
    main()
