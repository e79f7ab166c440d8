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
