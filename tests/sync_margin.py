"""How close the frame-sync decision is to a tie: the relative gap between the largest beta value and the largest
one in any OTHER column (only the column of the maximum is used, FrameSynchronisation.jl:66,76).  "Identical
frame-sync indices" between two implementations whose beta values differ at the 1e-7 level is a guarantee only
while this margin is well above that; tests and bench.py print it."""
import numpy as np


def beta_margin(beta):
    """beta: (W, n) matrix.  -> (1-based argmax column, relative margin to the best other column)"""
    colmax = np.max(np.asarray(beta, np.float64), axis=0)
    c = int(np.argmax(colmax))
    top = colmax[c]
    rest = np.delete(colmax, c)
    second = float(np.max(rest)) if rest.size else 0.0
    return c + 1, (top - second) / abs(top) if top != 0 else 0.0


def peak_margin_db(G, guard=2):
    """G: dB vector.  -> (0-based argmax, gap in dB to the largest value more than `guard` samples away)"""
    G = np.asarray(G, np.float64)
    i = int(np.argmax(G))
    m = np.ones(G.size, bool)
    m[max(0, i - guard): i + guard + 1] = False
    return i, float(G[i] - np.max(G[m])) if m.any() else float("inf")


def fast_vs_oracle(ctx, tsdr, O, iq, S, y_t, x_t, alpha, want_raster, rtol, guard=True, tie_tol=2e-6):
    """One buffer through tsdr_frames in TSDR_FAST mode and through the oracle, frame by frame.

    guard=True (the library default): the sync guard is on, and the bar is north_star's -- IDENTICAL frame-sync
    indices on every frame, no escape; rasters and frames (the IIR output) within rtol on every frame.

    guard=False ("sync_guard_ppb" = 0, kept only to show what the guard is for): an index may differ where the oracle's
    own beta values at the two columns differ by less than tie_tol (relative) -- the decision was a tie at the level of
    FAST's beta error (~3e-7), which a non-bit-exact evaluation cannot be asked to break the same way.  The criterion is
    applied per frame and indices keep being compared on every later frame (they do not depend on the IIR state);
    frames are compared up to the first divergent shift, rasters always.

    Returns dict(n_frames, ties=[(frame, axis, gpu, oracle, rel)], worst, guard=(checked, re-evaluated))."""
    z = np.ascontiguousarray(iq, np.complex64)
    nb = z.size // S
    gs = np.zeros((600, 800), np.float32, order="F")
    os_ = np.zeros((600, 800), np.float32, order="F")
    ctx.set_option("sync_guard_ppb", 20000 if guard else 0)
    ctx.sync_guard_stats(reset=True)
    try:
        g = ctx.frames(tsdr.SyncXY(ctx, 600, 800), z, S, y_t, x_t, np.float32(alpha), gs, want_raster=want_raster)
        gstats = ctx.sync_guard_stats()
    finally:
        ctx.set_option("sync_guard_ppb", 20000)
    assert g["n_frames"] == nb
    assert gstats[0] == (nb if guard else 0), gstats
    osync = O.SyncXY(600, 800)
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.maximum(np.abs(np.asarray(b, np.float64)), 1e-30)))
    ties, worst, diverged = [], 0.0, False
    prev_cy = None
    for f in range(nb):
        o = O.frames(osync, z[f * S:(f + 1) * S], S, y_t, x_t, np.float32(alpha), os_, want_raster=want_raster)
        cx = np.max(osync.beta("x").astype(np.float64), axis=0)
        cy = np.max(osync.beta("y").astype(np.float64), axis=0)
        gi, oi = [int(v) for v in g["sync_idx"][f]], [int(v) for v in o["sync_idx"][0]]
        if want_raster:
            worst = max(worst, rel(g["raster"][f], o["raster"][0]))
        if gi != oi:
            assert not guard, f"frame {f}: sync indices differ with the guard on: gpu {gi} oracle {oi}"
            for axis, cm in ((1, cx), (0, prev_cy)):
                if gi[axis] != oi[axis]:
                    assert cm is not None, f"frame {f}: first-frame s_y differs ({gi} vs {oi})"
                    d = abs(cm[gi[axis] - 1] - cm[oi[axis] - 1]) / abs(cm[oi[axis] - 1])
                    assert d < tie_tol, f"frame {f} axis {axis}: gpu {gi} oracle {oi}, beta differs by {d:.3e} (not a tie)"
                    ties.append((f, axis, gi[axis], oi[axis], d))
            diverged = True
        if not diverged:
            worst = max(worst, rel(g["frames"][f], o["frames"][0]))
        prev_cy = cy
    assert worst < rtol, worst
    if not diverged:
        assert rel(gs, os_) < rtol
    if ties:
        print(f"[sync_margin] guard off: {len(ties)} of {nb} frames used the tie escape: {ties}")
    return {"n_frames": nb, "ties": ties, "worst": worst, "guard": gstats}
