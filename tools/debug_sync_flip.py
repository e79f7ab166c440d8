"""Development aid (GPU box): where do FAST-mode sync indices differ from the oracle's, and how close to a tie was
the decision there?  Steps the oracle one frame at a time so that the beta matrices of every frame are visible."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from tempest_loader import load_package
T = load_package()
import importlib
synth = importlib.import_module("tempestsdr_jl_amd.synth")
import oracle_lib as O
from sync_margin import beta_margin
ctx = T.Context()
Fs, x_t, y_t, fv, nfr = 2.0e6, 1056, 628, 60.0, 66
S = synth.samples_per_frame(Fs, fv)
iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 11)
gs = np.zeros((600, 800), np.float32, order="F")
g = ctx.frames(T.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), gs)
os_ = np.zeros((600, 800), np.float32, order="F")
osync = O.SyncXY(600, 800)
prev_by = None
for f in range(nfr):
    o = O.frames(osync, iq[f * S:(f + 1) * S], S, y_t, x_t, np.float32(0.1), os_)
    bx, by = osync.beta("x"), osync.beta("y")
    gi, oi = g["sync_idx"][f], o["sync_idx"][0]
    if not np.array_equal(gi, oi):
        cx = np.max(bx.astype(np.float64), axis=0)
        print(f"frame {f}: gpu {gi.tolist()} oracle {oi.tolist()}")
        if gi[1] != oi[1]:
            print(f"   beta_x colmax at oracle col {cx[oi[1]-1]!r} at gpu col {cx[gi[1]-1]!r} rel diff {(cx[oi[1]-1]-cx[gi[1]-1])/cx[oi[1]-1]:.3e}")
        if gi[0] != oi[0] and prev_by is not None:
            cy = np.max(prev_by.astype(np.float64), axis=0)
            print(f"   beta_y(prev) colmax at oracle col {cy[oi[0]-1]!r} at gpu col {cy[gi[0]-1]!r} rel diff {(cy[oi[0]-1]-cy[gi[0]-1])/cy[oi[0]-1]:.3e}")
    prev_by = by
print("done; margins of last frame:", beta_margin(bx), beta_margin(by))
