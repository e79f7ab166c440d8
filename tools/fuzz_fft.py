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
PCT = int(sys.argv[2]) if len(sys.argv) > 2 else 100     # percentage of the case counts below (tests/test_fuzz_gpu.py runs a subset)
PCT = min(PCT, 100)
def cases(n): return max(1, n * PCT // 100)
ctx = T.Context()
def relmax(g, w):
    w = np.asarray(w); return float(np.max(np.abs(np.asarray(g, dtype=w.dtype) - w)) / np.max(np.abs(w)))
def smooth(maxn):
    while True:
        a, b, c = rng.integers(0, 23), rng.integers(0, 10), rng.integers(0, 8)
        n = (2 ** int(a)) * (3 ** int(b)) * (5 ** int(c))
        if 2 <= n <= maxn: return int(n)
worst = 0.0
for it in range(cases(60)):
    n = smooth(6_000_000) if it % 3 else int(rng.integers(2, 200_000))
    batch = 1 if n > 100_000 else int(rng.integers(1, 4))
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    for inv in (False, True):
        ref = (np.fft.ifft if inv else np.fft.fft)(x.astype(np.complex128), axis=1)
        e = relmax(ctx.fft(x if batch > 1 else x[0], inverse=inv), ref if batch > 1 else ref[0]); worst = max(worst, e)
        assert e < 1e-5, (n, batch, inv, e)
print(f"fft: {cases(60)} lengths ok, worst", worst)
worst = 0.0
for mixed in (0, 1):
    ctx.set_option("ac_mixed", mixed)
    for it in range(cases(12)):
        n = 2 * smooth(1_000_000) if it % 2 else int(rng.integers(1500, 500_000))
        x = (rng.random(n) ** 2).astype(np.float32) * 1e-5
        Fs = 1e6; maxd = (n // 2) / Fs
        mind = 0.0 if it % 3 == 0 else float(rng.integers(0, max(n // 6, 1))) / Fs      # indexMin > 1: the lag vector starts later
        if it % 4 == 1:   # a signal longer than 2*indexMax: only its first 2*indexMax samples are used (:27)
            x = np.concatenate([x, (rng.random(int(rng.integers(1, 5000))) ** 2).astype(np.float32)])
        g, _ = ctx.calculate_autocorrelation(x, Fs, mind, maxd, "lin")
        o, _ = O.calculate_autocorrelation(x, Fs, mind, maxd, "lin")
        e = relmax(g, o); worst = max(worst, e)
        assert g.shape == o.shape and e < 4e-5, (n, mixed, e)
print(f"autocorr: {2 * cases(12)} cases ok, worst", worst)

# the fused search (lags + zoom window + findmax in one call) on random lengths / windows, real and IQ input
ctx.set_option("ac_mixed", 1)
worst = 0.0
for it in range(cases(16)):
    n = 2 * smooth(2_500_000) if it % 4 else int(rng.integers(20_000, 600_000))
    if n < 20_000: n = 2 * 3 ** 9
    Fs = float(rng.choice([1e6, 2e6, 20e6]))
    cplx = bool(it % 2)
    t = np.arange(n)
    per = int(rng.integers(n // 12, n // 5))          # a periodic component so that the window holds a real peak
    base = (0.2 + (t % per < per // 7)).astype(np.float32)
    if cplx:
        x = (base * np.exp(2j * np.pi * rng.random(n))).astype(np.complex64) * 1e-2
        pw = (x.real.astype(np.float32) ** 2 + x.imag.astype(np.float32) ** 2)
    else:
        x = (base * (1 + 0.05 * rng.random(n))).astype(np.float32) * 1e-2
        pw = x
    maxd = (n // 2) / Fs
    rmin, rmax = float(rng.uniform(20, 60)), float(rng.uniform(70, 140))
    mind = 0.0 if it % 3 else float(rng.integers(1, 50)) / Fs
    G, pos, val = ctx.autocorr_search(x, Fs, mind, maxd, rmin, rmax, "log")
    o, _ = O.calculate_autocorrelation(pw if not cplx else ctx.abs2(x), Fs, mind, maxd, "log")
    assert G.shape == o.shape
    e = float(np.max(np.abs(G - o))); worst = max(worst, e)
    assert e < 5e-4, ("search dB", n, cplx, e)
    _, zw = ctx.zoom_autocorr(G, Fs, rmin, rmax)
    if zw.size:
        want = int(np.argmax(zw))                     # first maximum, like findmax
        assert pos == want and val == zw[want], ("search findmax", n, cplx, pos, want, val, float(zw[want]))
print(f"search: {cases(16)} cases ok, worst |dB| diff", worst)

# resampler!: random (bufferSize, upCoeff) incl. odd sizes (full-size route), 4096-point fast path, large primes
worst = 0.0
for it in range(cases(14)):
    up = int(rng.choice([1, 2, 3, 4, 5, 8]))
    nb = [int(rng.integers(4, 3000)), 4096 // up if 4096 % up == 0 else 1024, 2 * smooth(300_000), smooth(200_000) | 1,
          int(rng.integers(3000, 120_000))][it % 5]
    nb = max(nb, 4)
    r, ro = T.Resampler(ctx, nb, up), O.Resampler(nb, up)
    x = rng.standard_normal(nb).astype(np.float32)
    a, b = np.zeros(nb * up, np.float32), np.zeros(nb * up, np.float32)
    r(a, x); ro(b, x)
    e = float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)); worst = max(worst, e)
    assert e < 8e-6, ("resampler", nb, up, e)
    r.close()
print(f"resampler: {cases(14)} cases ok, worst", worst)
