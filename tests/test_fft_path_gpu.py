"""GPU parity, FFT-based functions: HIP (f32 FFT) vs the CPU oracle (f64 FFT rounded to f32).

These cannot be bit-exact (different FFT factorisation and precision), so the bar is a stated
tolerance: errors are measured relative to the largest magnitude of the vector, the natural
scale of FFT rounding noise; argmax-type results must be identical.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(7)

FFT_TOL = 5e-6   # max |err| / max |ref| for a single f32 FFT up to 2^23 points
CORR_TOL = 2e-5  # autocorrelation (two FFTs + squaring), relative to r[0]


def relmax(got, want):
    want = np.asarray(want)
    return float(np.max(np.abs(np.asarray(got, dtype=want.dtype) - want)) / np.max(np.abs(want)))


def crandn(n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


# ------------------------------------------------------------------ raw FFT engine
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 1 << 16, 1 << 17, 1 << 20, 1 << 22])
def test_fft_pow2(ctx, n):
    x = crandn(n)
    ref = np.fft.fft(x.astype(np.complex128))
    e = relmax(ctx.fft(x), ref)
    assert e < FFT_TOL, f"n={n}: {e:.3e}"
    ref = np.fft.ifft(x.astype(np.complex128))
    e = relmax(ctx.fft(x, inverse=True), ref)
    assert e < FFT_TOL, f"inverse n={n}: {e:.3e}"


@pytest.mark.parametrize("n,batch", [(8, 1000), (64, 33), (256, 17), (1024, 9), (4096, 3), (1 << 13, 2)])
def test_fft_batched(ctx, n, batch):
    x = crandn(n * batch).reshape(batch, n)
    ref = np.fft.fft(x.astype(np.complex128), axis=1)
    e = relmax(ctx.fft(x), ref)
    assert e < FFT_TOL, f"{e:.3e}"


@pytest.mark.parametrize("n", [3, 5, 7, 100, 997, 1000, 1536, 80_000, 4_000_000 // 64])
def test_fft_any_length(ctx, n):
    x = crandn(n)
    ref = np.fft.fft(x.astype(np.complex128))
    e = relmax(ctx.fft(x), ref)
    assert e < 2 * FFT_TOL, f"n={n}: {e:.3e}"
    e = relmax(ctx.fft(x, inverse=True), np.fft.ifft(x.astype(np.complex128)))
    assert e < 2 * FFT_TOL, f"inverse n={n}: {e:.3e}"
    # oracle FFT agrees with numpy too (pins the oracle's own transform)
    assert relmax(O.fft(x), ref) < 1e-12


@pytest.mark.parametrize("n,batch", [(512, 41), (1024, 37), (2048, 11), (4096, 5), (500, 33), (1000, 19), (2000, 6), (4000, 3), (768, 9),
                                     (1280, 7), (2500, 2), (3200, 3), (128, 300), (512, 2)])
def test_fft_rows_one_launch(ctx, n, batch):
    """Batched row transforms of 257 .. 4096 points at the lengths the three-step kernels serve (fft_mixed.hip:fft_rows_store;
    a ragged last tile: the batch is not a multiple of the rows per tile), both directions, against numpy."""
    x = crandn(n * batch).reshape(batch, n)
    assert relmax(ctx.fft(x), np.fft.fft(x.astype(np.complex128), axis=1)) < FFT_TOL
    assert relmax(ctx.fft(x, inverse=True), np.fft.ifft(x.astype(np.complex128), axis=1)) < FFT_TOL


# lengths 2^a 3^b 5^c go through the native mixed-radix passes: one pass (<= 256), two, three (2e6 = the packed
# half of the 4e6-sample search window), ragged tiles (odd first radix), radix-3 stages, batches
@pytest.mark.parametrize("n", [6, 10, 15, 25, 45, 120, 125, 200, 243, 250, 300, 625, 1000, 3000, 15625, 30000, 65610,
                               100_000, 390_625, 1_000_000, 2_000_000, 2_400_000, 60 * 150 * 120, 3_600_000])
def test_fft_mixed_radix(ctx, n):
    x = crandn(n)
    ref = np.fft.fft(x.astype(np.complex128))
    e = relmax(ctx.fft(x), ref)
    assert e < FFT_TOL, f"n={n}: {e:.3e}"
    e = relmax(ctx.fft(x, inverse=True), np.fft.ifft(x.astype(np.complex128)))
    assert e < FFT_TOL, f"inverse n={n}: {e:.3e}"


@pytest.mark.parametrize("n", [250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 80_000, 100_000, 3_000_000, 1500 * 2000, 20_000_000])
def test_fft_three_step_passes(ctx, n):
    """Factors of 500 / 1000 / 2000 (k_fft_mix3: three register steps, 8000-point tiles) as first, middle and last pass,
    ragged column tiles, and the same lengths with the option off (two-step kernels only) as the cross-check."""
    x = crandn(n)
    ref = np.fft.fft(x.astype(np.complex128))
    for big in (1, 0):
        ctx.set_option("fft_big", big)
        try:
            e = relmax(ctx.fft(x), ref)
            assert e < FFT_TOL, f"n={n} big={big}: {e:.3e}"
            e = relmax(ctx.fft(x, inverse=True), np.fft.ifft(x.astype(np.complex128)))
            assert e < FFT_TOL, f"inverse n={n} big={big}: {e:.3e}"
        finally:
            ctx.set_option("fft_big", 1)


@pytest.mark.parametrize("n", [1000, 30000, 390_625, 2_000_000, 2_400_000])
def test_fft_mixed_radix_lds_stage_kernel(ctx, n):
    """Factor sizes without a two-register-step kernel (e.g. 60, 120, 150) go through the generic LDS-stage kernel;
    the "fft_no_mix2" option sends every factor there so that it stays covered on the common sizes too."""
    ctx.set_option("fft_no_mix2", 1)
    try:
        x = crandn(n)
        assert relmax(ctx.fft(x), np.fft.fft(x.astype(np.complex128))) < FFT_TOL
        assert relmax(ctx.fft(x, inverse=True), np.fft.ifft(x.astype(np.complex128))) < FFT_TOL
    finally:
        ctx.set_option("fft_no_mix2", 0)


@pytest.mark.parametrize("n,batch", [(10, 777), (100, 41), (250, 300), (1000, 7), (6000, 5), (160_000, 2)])
def test_fft_mixed_radix_batched(ctx, n, batch):
    x = crandn(n * batch).reshape(batch, n)
    ref = np.fft.fft(x.astype(np.complex128), axis=1)
    assert relmax(ctx.fft(x), ref) < FFT_TOL
    assert relmax(ctx.fft(x, inverse=True), np.fft.ifft(x.astype(np.complex128), axis=1)) < FFT_TOL


# ------------------------------------------------------------------ Autocorrelations.jl
def test_autocorr_periodic_known_answer(ctx):
    """Exact period-T sequence: circular autocorrelation peaks at lag T, i.e. output index T
    (0-based) with minDelay=0; zoom labels that entry with rate Fs/(T+1) (reference off-by-one)."""
    T, reps, Fs = 250, 16, 10_000.0
    base = rng.random(T).astype(np.float32)
    x = np.tile(base, reps)  # n = 4000
    G, lags = ctx.calculate_autocorrelation(x, Fs, 0, 0.2)  # indexMax = 2000, n = 4000
    assert G.size == 2000 and lags.size == 2000 and lags[1] == 1 / Fs
    k = int(np.argmax(G[1:])) + 1
    assert k % T == 0, k
    assert abs(G[T] - G[0]) < 1e-3  # same energy at a full period (dB)
    rates, Gz = ctx.zoom_autocorr(G, Fs, rate_min=30, rate_max=50)
    # window is indices 200..333 (1-based); the peak sits at 1-based index T+1 = 251
    assert rates[0] == Fs / 200 and int(np.argmax(Gz)) + 200 == T + 1


@pytest.mark.parametrize("n,Fs,maxd,mind", [(3000, 30_000.0, 0.05, 0.0), (5000, 10_000.0, 0.1, 0.01), (4096, 4096.0, 0.5, 0.0),
                                           (1500, 1000.0, 1.0, 0.0), (100_003, 1e6, 0.03, 0.0),
                                           # n odd / n/2 with a factor 7: the zero-padded power-of-two route
                                           (4001, 1000.0, 4.0, 0.0), (14_000, 1000.0, 7.0, 0.5)])
@pytest.mark.parametrize("mixed", [False, True])
def test_autocorr_vs_oracle(ctx, n, Fs, maxd, mind, mixed):
    # "ac_mixed" (default on) routes n = 2 * (2^a 3^b 5^c) through the native mixed-radix transform (no padding, no
    # fold); off = the zero-padded power-of-two route, which also serves every other n
    ctx.set_option("ac_mixed", int(mixed))
    try:
        x = (rng.random(n) ** 2).astype(np.float32) * 1e-5  # power-like, non-negative (GUI.jl:70)
        for scale in ("lin", "log"):
            g, _ = ctx.calculate_autocorrelation(x, Fs, mind, maxd, scale)
            o, _ = O.calculate_autocorrelation(x, Fs, mind, maxd, scale)
            assert g.shape == o.shape
            if scale == "lin":
                assert relmax(g, o) < 2 * CORR_TOL, relmax(g, o)   # abs2 doubles the relative error
            else:
                assert np.max(np.abs(g - o)) < 2e-4, np.max(np.abs(g - o))  # dB; 8.7*CORR_TOL
    finally:
        ctx.set_option("ac_mixed", 1)


@pytest.mark.parametrize("n,Fs", [(80_000, 1e5), (200_000, 1e6), (180_000, 1e6), (60_000, 1e5), (3000, 30_000.0), (1_000_000, 5e6),
                                  (8192, 4096.0), (100_003, 1e6)])
def test_autocorr_fused_middle_and_fused_findmax(ctx, n, Fs):
    """The autocorrelation with its last forward pass, power spectrum and first inverse pass in one launch (k_fft_mid:
    two-pass splits, odd column counts, ragged tiles) against the two-transform route and the oracle; and
    tsdr_autocorr_search_d's findmax -- found by the launch that writes the lags -- against numpy on the lags it
    returned, on IQ input (abs2 formed while loading).  The power-of-two and zero-padded lengths take the routes
    without a fused middle / epilogue and must answer the same way."""
    z = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 3e-3).astype(np.complex64)
    z *= (1.0 + 0.5 * np.cos(2 * np.pi * np.arange(n) / 977.0)).astype(np.float32)   # a period for the zoom window to find
    x = O.abs2(z)
    maxd = (n // 2) / Fs
    o, _ = O.calculate_autocorrelation(x, Fs, 0, maxd)
    res = {}
    for fuse in (1, 0):
        ctx.set_option("ac_fuse_mid", fuse)
        try:
            g, _ = ctx.calculate_autocorrelation(x, Fs, 0, maxd)
            G, pos, val = ctx.autocorr_search(z, Fs, 0, maxd, rate_min=Fs / 3000, rate_max=Fs / 300)
        finally:
            ctx.set_option("ac_fuse_mid", 1)
        assert g.shape == o.shape == G.shape
        assert np.max(np.abs(g - o)) < 2e-4, (fuse, np.max(np.abs(g - o)))
        assert np.max(np.abs(G - o)) < 2e-4, (fuse, np.max(np.abs(G - o)))
        lo, hi = int(round(Fs / (Fs / 300))), min(int(round(Fs / (Fs / 3000))), G.size)   # zoom_autocorr bounds, 1-based
        win = G[lo - 1: hi]
        assert pos == int(np.argmax(win)) and val == win[pos], (fuse, pos, int(np.argmax(win)))
        res[fuse] = G
    assert np.max(np.abs(res[0] - res[1])) < 2e-4


def test_findmax_routes_interleaved_on_one_context(ctx):
    """Searches whose findmax is an FFT-pass epilogue (n = 2 * 2^a3^b5^c) and searches that run the separate argmax kernel
    (every other n), alternating on one context with the LARGER maxima first: the argmax kernel's two key slots change roles
    only when it runs (a stale key from an earlier search used to win the later one's atomicMax)."""
    Fs = 1e6
    def case(n, amp):
        z = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * amp).astype(np.complex64)
        z *= (1.0 + 0.5 * np.cos(2 * np.pi * np.arange(n) / 977.0)).astype(np.float32)
        return z
    order = [(100_003, 3e-1), (200_000, 3e-2), (100_003, 3e-3), (131_072, 3e-1), (180_000, 3e-2), (99_999, 3e-4), (65_536, 3e-4)]
    for n, amp in order:
        z = case(n, amp)
        G, pos, val = ctx.autocorr_search(z, Fs, 0, (n // 2) / Fs, rate_min=Fs / 3000, rate_max=Fs / 300)
        win = G[300 - 1: min(3000, G.size)]
        assert pos == int(np.argmax(win)) and val == win[pos], (n, amp, pos, int(np.argmax(win)), val, float(win.max()))


def test_autocorr_bounds_error(ctx):
    x = np.ones(100, np.float32)
    with pytest.raises(IndexError):  # BoundsError at Autocorrelations.jl:33
        ctx.calculate_autocorrelation(x, 1000.0, 0, 0.5)
    with pytest.raises(IndexError):
        O.calculate_autocorrelation(x, 1000.0, 0, 0.5)


def test_autocorr_search_c2(ctx, synth):
    """extract_configuration (GUI.jl:56-81) at the C2 size: n = 4e6, refresh-rate argmax."""
    Fs, fv = 20e6, 60.0
    iq = synth.synth_leak(Fs, 2576, 1125, fv, 4_000_000)
    x = O.abs2(iq)
    assert np.array_equal(ctx.abs2(iq).view(np.uint32), x.view(np.uint32))
    g, _ = ctx.calculate_autocorrelation(x, Fs, 0, 0.1)
    o, _ = O.calculate_autocorrelation(x, Fs, 0, 0.1)
    assert g.size == o.size == 2_000_000
    assert np.max(np.abs(g - o)) < 2e-4, np.max(np.abs(g - o))
    rg, gz = ctx.zoom_autocorr(g, Fs, rate_min=50, rate_max=90)
    ro, oz = O.zoom_autocorr(o, Fs, rate_min=50, rate_max=90)
    assert gz.size == oz.size == 177_779 and np.array_equal(rg, ro)
    pg, po = int(np.argmax(gz)), int(np.argmax(oz))
    top2 = np.sort(oz)[-2:]
    assert pg == po, f"argmax {pg} vs {po}; oracle top-2 margin {top2[1] - top2[0]:.3e} dB"
    # the strongest lag may sit one video line (Fs/(y_t*fv) = 296 samples = 0.053 Hz) off the frame
    # lag: the bar pattern is line-periodic and sub-sample alignment decides between the two
    assert abs(rg[pg] - fv) < 0.1, rg[pg]


def test_autocorr_partial_sums_to_whole(ctx):
    """partial sums over ranges of m, added in the linear domain, equal the whole."""
    n, n_lags, G = 40_000, 20_000, 4
    x = (rng.random(n) ** 2).astype(np.float32)
    dx = ctx.upload(x)
    parts = []
    for g in range(G):
        m0, cnt = g * n // G, n // G
        dp = ctx.dev_alloc(n_lags * 4)
        ctx.call("tsdr_autocorr_partial_d", C.c_void_p(dx), 0, n, m0, cnt, n_lags, C.c_void_p(dp))
        parts.append(ctx.download(dp, n_lags, np.float32))
        ctx.dev_free(dp)
    total = np.sum(np.stack(parts).astype(np.float64), axis=0)
    X = np.fft.fft(x.astype(np.float64))
    ref = np.fft.ifft(X * np.conj(X)).real[:n_lags]
    assert relmax(total, ref) < CORR_TOL, relmax(total, ref)
    # finish: 10log10(abs2) of the reduced vector
    dt = ctx.upload(total.astype(np.float32))
    do = ctx.dev_alloc(n_lags * 4)
    ctx.call("tsdr_autocorr_finish_d", C.c_void_p(dt), 0, n_lags, 1, C.c_void_p(do))
    db = ctx.download(do, n_lags, np.float32)
    assert np.max(np.abs(db - 20 * np.log10(np.abs(ref)))) < 2e-4
    idx, val = C.c_size_t(0), C.c_float(0)
    ctx.call("tsdr_argmax_d", C.c_void_p(do), n_lags, C.byref(idx), C.byref(val))
    assert idx.value == int(np.argmax(db)) and val.value == db[idx.value]
    for p in (dx, dt, do):
        ctx.dev_free(p)


# ------------------------------------------------------------------ GetSpectrum.jl
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("N", [1024, 80_000, 4097])
def test_spectrum_vs_oracle(ctx, N, cplx):
    sig = crandn(N + 10) if cplx else rng.standard_normal(N + 10).astype(np.float32)
    sig = sig + 3.0  # strong DC line, like an AM-demodulated capture
    f, y = ctx.getSpectrum(2e6, sig, N=N, lin=True)
    o = O.getSpectrum(sig, N=N, lin=True)
    assert f[0] == -1e6 and f.size == N
    assert relmax(np.sqrt(y), np.sqrt(o)) < 2 * FFT_TOL, relmax(np.sqrt(y), np.sqrt(o))
    # Parseval: sum |X|^2 = N * sum |x|^2
    assert abs(y.astype(np.float64).sum() / (N * np.sum(np.abs(sig[:N].astype(np.complex128)) ** 2)) - 1) < 1e-5
    # dB output where the bins are not in the noise of the DC line
    _, ydb = ctx.getSpectrum(2e6, sig, N=N)
    odb = O.getSpectrum(sig, N=N)
    strong = o > 1e-4 * o.max()
    assert np.max(np.abs(ydb[strong] - odb[strong])) < 2e-2


def test_spectrum_too_long_raises(ctx):
    with pytest.raises(IndexError):
        ctx.getSpectrum(1.0, np.ones(10, np.float32), N=11)


@pytest.mark.parametrize("sizeFFT,cplx", [(1024, False), (1024, True), (256, True), (1000, False), (1000, True), (2, True), (6, False),
                                          (4096, True), (4096, False), (3000, True), (2048, False), (960, True), (17, True), (2000, True), (2000, False), (500, True),
                                          (8192, True),
                                          # the other segment lengths with a three-step accumulator kernel (fft_mixed.hip:kWelch3)
                                          (128, True), (512, False), (512, True), (2048, True), (768, True), (1200, False), (1280, True), (1600, True),
                                          (2500, True), (3200, False), (4000, True)])
def test_welch_and_waterfall(ctx, sizeFFT, cplx):
    L = sizeFFT * 37 + min(123, sizeFFT - 1)  # ragged tail is dropped
    sig = crandn(L) if cplx else rng.standard_normal(L).astype(np.float32)
    _, y = ctx.getWelch(1.0, sig, sizeFFT=sizeFFT, lin=True)
    o = O.getWelch(sig, sizeFFT=sizeFFT, lin=True)
    assert relmax(y, o) < 4 * FFT_TOL, relmax(y, o)
    t, f, m = ctx.getWaterfall(1.0, sig, sizeFFT=sizeFFT)
    om = O.getWaterfall(sig, sizeFFT=sizeFFT)
    assert m.dtype == np.float64 and m.shape == (sizeFFT, 37) and m.flags.f_contiguous
    assert relmax(np.sqrt(m), np.sqrt(om)) < 4 * FFT_TOL
    assert t[1] == sizeFFT / 1.0


def test_welch_general_size_on_chip_many_segments(ctx):
    """getWelch at sizeFFT = 1000 over a whole C2 buffer (10 000 segments): the general 2^a 3^b 5^c route whose segment
    spectra never leave the chip (every workgroup accumulates abs2 of the transforms it forms, one partial sum each) against
    the oracle; real input too."""
    L = 10_000_000
    sig = crandn(L)
    _, y = ctx.getWelch(20e6, sig, sizeFFT=1000, lin=True)
    o = O.getWelch(sig, sizeFFT=1000, lin=True)
    assert relmax(y, o) < 4 * FFT_TOL, relmax(y, o)
    r = np.ascontiguousarray(sig.real[: 3000 * 1777 + 5])
    _, yr = ctx.getWelch(20e6, r, sizeFFT=3000, lin=True)
    assert relmax(yr, O.getWelch(r, sizeFFT=3000, lin=True)) < 4 * FFT_TOL


def test_welch_waterfall_c2_buffer(ctx):
    """getWelch / getWaterfall over one C2 capture buffer (1e7 complex samples = 9765 segments of 1024: the LDS-resident
    one-wavefront-per-segment path with its blocked sum over segments) against the oracle's f64 FFTs and strictly
    sequential f32 sum."""
    L = 10_000_000
    sig = crandn(L)
    _, y = ctx.getWelch(20e6, sig, lin=True)
    o = O.getWelch(sig, lin=True)
    assert relmax(y, o) < 4 * FFT_TOL, relmax(y, o)
    _, ydb = ctx.getWelch(20e6, sig)
    assert np.max(np.abs(ydb - O.getWelch(sig))) < 1e-3
    t, f, m = ctx.getWaterfall(20e6, sig[: 1024 * 3000 + 17])
    om = O.getWaterfall(sig[: 1024 * 3000 + 17])
    assert m.shape == (1024, 3000) and relmax(np.sqrt(m), np.sqrt(om)) < 4 * FFT_TOL


# ------------------------------------------------------------------ init_resampler
@pytest.mark.parametrize("n", [1, 2, 7, 64, 97, 1000, 2997, 4096, 30030, 100003])
def test_fft64(ctx, n):
    """the device f64 transform behind initLPF (Stockham passes for 2/3/5/7/11/13-smooth lengths, Bluestein otherwise)
    against numpy's f64 FFT"""
    z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    for inv in (False, True):
        want = np.fft.ifft(z) if inv else np.fft.fft(z)
        got = ctx.fft64(z, inverse=inv)
        assert np.max(np.abs(got - want)) < 1e-12 * np.max(np.abs(want)) * max(1.0, np.log2(n + 1)), n


# (1000, 4) / (100000, 5): even bufferSize -> the half-size route (forward transform of bufferSize/2 points on the input as
# it lies, one pointwise kernel, inverse transform of sizeFFT/2 points straight into `out`); (999, 3) / (625, 3): odd
# bufferSize -> full-size transforms with fused loaders / Bluestein; (64, 2): one pass; sizeFFT = 4096: one workgroup
@pytest.mark.parametrize("bufferSize,up", [(1000, 4), (1024, 2), (999, 3), (64, 2), (4096, 8), (100000, 5), (625, 3), (1024, 4), (512, 8),
                                           (2048, 2), (10, 1), (4, 2), (250000, 4), (3000, 7),
                                           # sizeFFT divisible by 6: round.(exp(im*theta)) has an entry 0.5 -/+ 1e-13 away from a tie, decided by the
                                           # last bit of the range element 2pi*k/sizeFFT (the library once formed it with the long-double pi: 0.9 % off)
                                           (6, 1), (1218, 2), (2691, 1), (6561, 2), (3000, 2)])
def test_init_resampler(ctx, bufferSize, up):
    r, o = ctx.init_resampler(np.float32, bufferSize, up), O.Resampler(bufferSize, up)
    # H is ComplexF64 on both sides (Resampler.jl:93-97): the device builds it with its own f64 transforms
    H64, Ho = r.lpf64(), o.lpf()
    assert np.max(np.abs(H64 - Ho)) < 2e-7 * np.max(np.abs(Ho)), np.max(np.abs(H64 - Ho)) / np.max(np.abs(Ho))
    assert relmax(r.lpf(), Ho) < 2e-7
    x = rng.standard_normal(bufferSize).astype(np.float32)
    out, oo = np.empty(bufferSize * up, np.float32), np.empty(bufferSize * up, np.float32)
    r(out, x)
    o(oo, x)
    assert relmax(out, oo) < 4e-6, relmax(out, oo)   # two f32 transforms against the oracle's f64 ones
    with pytest.raises(AssertionError):  # Resampler.jl:47
        r(out, x[:-1])
    with pytest.raises(AssertionError):  # Resampler.jl:44
        r(out.astype(np.float64), x.astype(np.float64))
