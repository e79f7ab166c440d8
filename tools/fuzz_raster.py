"""Development aid (GPU box): random geometries through FAST sig_to_image and the frame path vs the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from tempest_loader import load_package
T = load_package()
import oracle_lib as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
PCT = int(sys.argv[2]) if len(sys.argv) > 2 else 100     # percentage of the case counts below (tests/test_fuzz_gpu.py runs a subset)
def cases(n): return max(1, n * PCT // 100)
ctx = T.Context()
def relerr(g, w):
    w = np.asarray(w, np.float64); return float(np.max(np.abs(np.asarray(g, np.float64) - w) / np.maximum(np.abs(w), 1e-30)))
worst = 0.0
for it in range(cases(40)):
    y_t = int(rng.integers(20, 1400)); x_t = int(rng.integers(30, 3000))
    ratio = float(np.exp(rng.uniform(np.log(0.05), np.log(3.0))))
    S = max(2, int(y_t * x_t * ratio))
    sig = (0.05 + rng.random(S, dtype=np.float32))
    e = relerr(ctx.sig_to_image(sig, y_t, x_t), O.sig_to_image(sig, y_t, x_t))
    worst = max(worst, e)
    assert e < 1e-6, (S, y_t, x_t, e)
print(f"sig_to_image: {cases(40)} geometries ok, worst", worst)
worst = 0.0
for it in range(cases(16)):
    y_t = int(rng.integers(130, 1300)); x_t = int(rng.integers(260, 2800)); nfr = int(rng.integers(1, 4))
    ratio = float(np.exp(rng.uniform(np.log(0.08), np.log(1.6))))
    S = max(2, int(y_t * x_t * ratio))
    iq = ((rng.standard_normal(S * nfr + 3) + 1j * rng.standard_normal(S * nfr + 3)) * 1e-3).astype(np.complex64)
    gs = np.zeros((600, 800), np.float32, order="F"); os_ = np.zeros((600, 800), np.float32, order="F")
    for want_raster in (True, False):
        gs[:] = 0; os_[:] = 0
        g = ctx.frames(T.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), gs, want_raster=want_raster)
        o = O.frames(O.SyncXY(600, 800), iq, S, y_t, x_t, np.float32(0.1), os_, want_raster=want_raster)
        assert np.array_equal(g["sync_idx"], o["sync_idx"]), (S, y_t, x_t, g["sync_idx"].tolist(), o["sync_idx"].tolist())
        for f in range(nfr):
            e = relerr(g["frames"][f], o["frames"][f]); worst = max(worst, e)
            assert e < 1e-6, (S, y_t, x_t, f, e)
            if want_raster:
                e = relerr(g["raster"][f], o["raster"][f]); worst = max(worst, e)
                assert e < 1e-6, (S, y_t, x_t, f, e)
print(f"frames: {cases(16)} geometries x 2 ok, worst", worst)

# EXACT mode: bit-identical rasters, frames, state and indices on random geometries
ctx.set_precision("exact")
beq = lambda a, b: np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
for it in range(cases(10)):
    y_t = int(rng.integers(20, 1300)); x_t = int(rng.integers(30, 2800)); nfr = int(rng.integers(1, 3))
    ratio = float(np.exp(rng.uniform(np.log(0.08), np.log(1.6))))
    S = max(2, int(y_t * x_t * ratio))
    iq = ((rng.standard_normal(S * nfr + 3) + 1j * rng.standard_normal(S * nfr + 3)) * 1e-3).astype(np.complex64)
    gs = np.zeros((600, 800), np.float32, order="F"); os_ = np.zeros((600, 800), np.float32, order="F")
    for want_raster in (True, False):   # (False: the raster-free kernel, k_down_fused<EXACT>)
        gs[:] = 0; os_[:] = 0
        g = ctx.frames(T.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), gs, want_raster=want_raster)
        o = O.frames(O.SyncXY(600, 800), iq, S, y_t, x_t, np.float32(0.1), os_, want_raster=want_raster)
        assert np.array_equal(g["sync_idx"], o["sync_idx"]) and beq(gs, os_), (S, y_t, x_t, want_raster)
        for f in range(nfr):
            assert beq(g["frames"][f], o["frames"][f]), (S, y_t, x_t, f, want_raster)
            assert not want_raster or beq(g["raster"][f], o["raster"][f]), (S, y_t, x_t, f)
print(f"exact frames: {cases(10)} geometries x 2 bit-identical")
ctx.set_precision("fast")
