"""Development aid (GPU box): random lengths through tsdr_fft_c2c (power of two / 2^a3^b5^c / Bluestein) and the
autocorrelation routes vs numpy / the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from tempest_loader import load_package
T = load_package()
import oracle_lib as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = T.Context()
def relmax(g, w):
    w = np.asarray(w); return float(np.max(np.abs(np.asarray(g, dtype=w.dtype) - w)) / np.max(np.abs(w)))
def smooth(maxn):
    while True:
        a, b, c = rng.integers(0, 23), rng.integers(0, 10), rng.integers(0, 8)
        n = (2 ** int(a)) * (3 ** int(b)) * (5 ** int(c))
        if 2 <= n <= maxn: return int(n)
worst = 0.0
for it in range(60):
    n = smooth(6_000_000) if it % 3 else int(rng.integers(2, 200_000))
    batch = 1 if n > 100_000 else int(rng.integers(1, 4))
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    for inv in (False, True):
        ref = (np.fft.ifft if inv else np.fft.fft)(x.astype(np.complex128), axis=1)
        e = relmax(ctx.fft(x if batch > 1 else x[0], inverse=inv), ref if batch > 1 else ref[0]); worst = max(worst, e)
        assert e < 1e-5, (n, batch, inv, e)
print("fft: 60 lengths ok, worst", worst)
worst = 0.0
for mixed in (0, 1):
    ctx.set_option("ac_mixed", mixed)
    for it in range(12):
        n = 2 * smooth(1_000_000) if it % 2 else int(rng.integers(1500, 500_000))
        x = (rng.random(n) ** 2).astype(np.float32) * 1e-5
        Fs = 1e6; maxd = (n // 2) / Fs
        g, _ = ctx.calculate_autocorrelation(x, Fs, 0.0, maxd, "lin")
        o, _ = O.calculate_autocorrelation(x, Fs, 0.0, maxd, "lin")
        e = relmax(g, o); worst = max(worst, e)
        assert g.shape == o.shape and e < 4e-5, (n, mixed, e)
print("autocorr: 24 cases ok, worst", worst)
