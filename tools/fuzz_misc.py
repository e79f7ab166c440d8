"""Development aid (GPU box): random sizes through imresize (1-D / 2-D), SyncXY.vsync and getSpectrum / getWelch
vs the CPU oracle.  EXACT mode must stay bit-identical on the resize / vsync rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from tempest_loader import load_package
T = load_package()
import oracle_lib as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = T.Context()
beq = lambda a, b: np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))
ctx.set_precision("exact")
for it in range(30):
    n_in, n_out = int(rng.integers(2, 300_000)), int(rng.integers(1, 300_000))
    x = rng.standard_normal(n_in).astype(np.float32)
    assert beq(ctx.imresize1d(x, n_out), O.imresize1d(x, n_out)), ("resize1d", n_in, n_out)
for it in range(30):
    h, w, ho, wo = (int(rng.integers(2, 900)) for _ in range(4))
    img = np.asfortranarray(rng.random((h, w), dtype=np.float32))
    assert beq(ctx.imresize2d(img, (ho, wo)), O.imresize2d(img, (ho, wo))), ("resize2d", h, w, ho, wo)
print("resize: 60 cases bit-identical")
for it in range(12):
    y, x = int(rng.integers(40, 700)), int(rng.integers(40, 900))
    sg, so = T.SyncXY(ctx, y, x), O.SyncXY(y, x)
    for rep in range(3):
        img = np.asfortranarray(rng.random((y, x), dtype=np.float32))
        if rep == 1:  # a blank band like a real frame
            img[:, x // 3: x // 3 + x // 7] *= 0.05; img[y // 2: y // 2 + y // 15, :] *= 0.05
        a, b = sg.vsync(img), so.vsync(img)
        assert tuple(a) == tuple(b), ("vsync", y, x, rep, a, b)
print("vsync: 36 calls identical")
ctx.set_precision("fast")
worst = 0.0
for it in range(20):
    N = int(rng.integers(16, 400_000)); cplx = bool(it % 2)
    sig = (rng.standard_normal(N) + (1j * rng.standard_normal(N) if cplx else 0)).astype(np.complex64 if cplx else np.float32)
    f, y = ctx.getSpectrum(1e6, sig, N, lin=True); o = O.getSpectrum(sig, N, lin=True)
    e = float(np.max(np.abs(np.sqrt(y) - np.sqrt(o))) / np.max(np.sqrt(o))); worst = max(worst, e)
    assert e < 2e-5, ("spectrum", N, cplx, e)
for it in range(10):
    sz = int(2 ** rng.integers(4, 13)) if it % 2 else int(rng.integers(16, 3000)); nseg = int(rng.integers(1, 40))
    sig = (rng.standard_normal(sz * nseg + 5) + 1j * rng.standard_normal(sz * nseg + 5)).astype(np.complex64)
    f, y = ctx.getWelch(1e6, sig, sz, lin=True); o = O.getWelch(sig, sz, lin=True)
    e = float(np.max(np.abs(y - o)) / np.max(o)); worst = max(worst, e)
    assert e < 4e-5, ("welch", sz, nseg, e)
print("spectrum/welch: 30 cases ok, worst", worst)
