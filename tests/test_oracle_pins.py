"""Pins for the CPU oracle (oracle/tempest_oracle.c).  CPU only.

The reference has no golden vectors for this path and cannot be executed here (no Julia), so
the oracle is pinned three ways:
  1. analytic known answers that need no oracle at all,
  2. independent second implementations of the same published conventions available in this
     container (numpy FFT, torch interpolate align_corners=False, scipy lfilter / blackman),
  3. an independently written numpy twin of the resize rule, required to agree BIT FOR BIT.
Agreement with (2) shows the restatement follows those well-known conventions, not that it
matches Julia: parity with the Julia packages stays UNPINNED.
"""
import numpy as np
import pytest

import oracle_lib as O

rng = np.random.default_rng(1)


# ------------------------------------------------------------------ Demodulation.jl
def test_amdemod_unit_circle_and_ulp():
    phi = rng.uniform(0, 2 * np.pi, 10000)
    z = np.exp(1j * phi).astype(np.complex64)
    a = O.amDemod(z)
    assert np.max(np.abs(a - 1.0)) <= 2 ** -23  # |e^{j phi}| = 1 to one ulp (the inputs are rounded)
    z = (rng.standard_normal(100000) + 1j * rng.standard_normal(100000)).astype(np.complex64) * 3e-3
    ref = np.hypot(z.real.astype(np.float64), z.imag.astype(np.float64))
    assert np.max(np.abs(O.amDemod(z) - ref) / ref) <= 2 ** -24 * 1.0001  # correctly rounded
    assert O.amDemod(np.array([complex(np.inf, np.nan)], np.complex64))[0] == np.inf  # hypot(Inf,NaN)=Inf


def test_invert_amdemod_and_fmdemod():
    z = (rng.standard_normal(1000) + 1j * rng.standard_normal(1000)).astype(np.complex64)
    a = np.abs(z.astype(np.complex128))
    got = O.invert_amDemod(z)
    assert np.allclose(got, 1 - a / a.max(), atol=2e-7)
    assert got.min() == 0.0
    with pytest.raises(AssertionError):
        O.invert_amDemod(np.zeros(0, np.complex64))
    # constant-frequency tone: phase increment is the angle everywhere, out[0] = 0
    w = 0.3
    z = np.exp(1j * w * np.arange(500)).astype(np.complex64)
    f = O.fmDemod(z)
    assert f[0] == 0 and np.allclose(f[1:], w, atol=1e-6)


# ------------------------------------------------------------------ imresize restatement
def np_imresize1d(x, n_out):
    """independent numpy restatement of ImageTransformations.imresize! + Linear B-spline"""
    x = np.asarray(x, np.float32)
    n_in = x.size
    if n_in == n_out:
        return x.copy()
    sf = np.float64(n_in) / np.float64(n_out)
    off = (1.0 - 0.5) - sf * (1.0 - 0.5)
    i = np.arange(1, n_out + 1, dtype=np.float64)
    c = np.clip(sf * i + off, 1.0, float(n_in))
    xf = np.floor(c)
    xf = np.where(xf > n_in - 1, xf - 1, xf)
    d = c - xf
    k = xf.astype(np.int64) - 1
    return ((1.0 - d) * x[k].astype(np.float64) + d * x[k + 1].astype(np.float64)).astype(np.float32)


@pytest.mark.parametrize("n_in,n_out", [(100, 873), (1000, 37), (64, 64), (2, 5), (333333, 28980), (4999, 5000)])
def test_resize1d_numpy_twin_bitexact(n_in, n_out):
    x = rng.random(n_in, dtype=np.float32)
    assert np.array_equal(O.imresize1d(x, n_out).view(np.uint32), np_imresize1d(x, n_out).view(np.uint32))


def test_resize_ramp_is_ramp():
    # linear interpolation reproduces a linear ramp exactly in the interior (edges clamp)
    x = np.arange(1000, dtype=np.float32)
    y = O.imresize1d(x, 4000)
    expected = (np.arange(4000) + 0.5) * 0.25 - 0.5
    interior = (expected >= 0) & (expected <= 999)
    assert np.max(np.abs(y[interior] - expected[interior])) < 1e-4
    assert y[0] == 0.0 and y[-1] == 999.0  # clamped


def test_resize_matches_align_corners_false_convention():
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    x = rng.random(257, dtype=np.float32)
    for n_out in (1000, 64):
        t = F.interpolate(torch.tensor(x)[None, None], size=n_out, mode="linear", align_corners=False)[0, 0].numpy()
        assert np.max(np.abs(O.imresize1d(x, n_out) - t)) < 3e-5  # torch computes the source coordinate in f32 (error ~ n_in * 2^-24 * slope)
    img = np.asfortranarray(rng.random((45, 64), dtype=np.float32))
    for size in ((20, 30), (100, 90)):
        t = F.interpolate(torch.tensor(np.ascontiguousarray(img))[None, None], size=size, mode="bilinear",
                          align_corners=False)[0, 0].numpy()
        assert np.max(np.abs(O.imresize2d(img, size) - t)) < 3e-5


def test_resize2d_separable_against_1d():
    # 2-D linear resize = 1-D along each axis (in f64 before the single final rounding)
    img = np.asfortranarray(rng.random((31, 47), dtype=np.float32))
    got = O.imresize2d(img, (12, 90))
    rows = np.stack([np_imresize1d(img[:, c], 12).astype(np.float64) for c in range(47)], axis=1)
    ref = np.stack([np_imresize1d(rows[r, :].astype(np.float32), 90) for r in range(12)], axis=0)
    assert np.max(np.abs(got - ref)) < 1e-6


def test_sig_to_image_layout():
    # element (l,p) of the (y_t,x_t) matrix is resized[l*x_t + p]   (Resampler.jl:117-122)
    y_t, x_t = 7, 11
    sig = rng.random(50, dtype=np.float32)
    img = O.sig_to_image(sig, y_t, x_t)
    flat = O.imresize1d(sig, y_t * x_t)
    assert img.shape == (y_t, x_t)
    for l in range(y_t):
        assert np.array_equal(img[l, :], flat[l * x_t:(l + 1) * x_t])
    # copy path when S == y_t*x_t
    sig = rng.random(77, dtype=np.float32)
    assert np.array_equal(O.sig_to_image(sig, 7, 11), sig.reshape(7, 11))
    assert O.downgradeImage(np.asfortranarray(rng.random((40, 50), dtype=np.float32))).shape == (600, 800)
    assert np.array_equal(O.naiveResampler(np.array([1, 2, 3], np.float32), 2), np.array([1, 1, 2, 2, 3, 3], np.float32))


# ------------------------------------------------------------------ FFT / Autocorrelations.jl
@pytest.mark.parametrize("n", [1, 2, 8, 12, 100, 125, 128, 997, 1000, 4096, 15625])
def test_oracle_fft_vs_numpy(n):
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert np.allclose(O.fft(x), np.fft.fft(x), rtol=0, atol=1e-10 * n)
    assert np.allclose(O.fft(x, inverse=True), np.fft.ifft(x), rtol=0, atol=1e-12)


def test_autocorr_is_circular_unnormalised():
    x = rng.random(64).astype(np.float32)
    g, lags = O.calculate_autocorrelation(x, 64.0, 0, 0.5, "lin")  # indexMax=32, n=64
    xd = x.astype(np.float64)
    ref = np.array([np.sum(xd * np.roll(xd, -k)) for k in range(32)]) ** 2
    assert g.size == 32 and np.allclose(g, ref, rtol=1e-6)
    assert lags[0] == 0 and lags[1] == 1 / 64.0
    # n = min(2*indexMax, length(x)): a shorter window than the signal
    g2, _ = O.calculate_autocorrelation(rng.random(1000).astype(np.float32), 100.0, 0.05, 0.2)  # n = 40
    assert g2.size == 20 - 6 + 1
    with pytest.raises(IndexError):
        O.calculate_autocorrelation(np.ones(10, np.float32), 100.0, 0, 0.5)


def test_autocorr_periodic_peak_and_zoom_off_by_one():
    T, Fs = 125, 5000.0
    x = np.tile(rng.random(T).astype(np.float32), 8)  # n = 1000
    G, _ = O.calculate_autocorrelation(x, Fs, 0, 0.1)  # 500 lags, n = 1000
    assert int(np.argmax(G[1:])) + 1 in (T, 2 * T, 3 * T)
    assert abs(G[T] - G[0]) < 1e-4  # a full period has the zero-lag energy
    # zoom window [round(Fs/rate_max), round(Fs/rate_min)] is 1-based; entry k is lag k-1 but labelled k/Fs
    pmin, pmax = O.zoom_bounds(500, Fs, 30, 50)
    assert (pmin, pmax) == (100, 167)
    rates, Gz = O.zoom_autocorr(G, Fs, 30, 50)
    assert rates[0] == Fs / 100 and Gz[0] == G[99]
    assert int(np.argmax(Gz)) + pmin == T + 1  # lag T lives at 1-based index T+1


# ------------------------------------------------------------------ GetSpectrum.jl
def test_spectrum_parseval_and_shift():
    N = 1000
    x = rng.standard_normal(N).astype(np.float32)
    y = O.getSpectrum(x, lin=True)
    assert abs(y.astype(np.float64).sum() / (N * np.sum(x.astype(np.float64) ** 2)) - 1) < 1e-5
    ref = np.abs(np.fft.fftshift(np.fft.fft(x.astype(np.float64)))) ** 2
    assert np.allclose(y, ref, rtol=1e-4, atol=1e-3)
    x = rng.standard_normal(7).astype(np.float32)  # odd N: fftshift moves ceil(N/2)
    assert np.allclose(O.getSpectrum(x, lin=True), np.abs(np.fft.fftshift(np.fft.fft(x.astype(np.float64)))) ** 2, rtol=1e-4)
    assert np.allclose(O.getSpectrum(x), 10 * np.log10(np.abs(np.fft.fftshift(np.fft.fft(x.astype(np.float64)))) ** 2), atol=1e-3)


def test_welch_is_sum_not_mean_and_waterfall():
    sz, nb = 64, 9
    x = (rng.standard_normal(sz * nb + 5) + 1j * rng.standard_normal(sz * nb + 5)).astype(np.complex64)
    seg = x[: sz * nb].reshape(nb, sz).astype(np.complex128)
    P = np.abs(np.fft.fft(seg, axis=1)) ** 2
    assert np.allclose(O.getWelch(x, sz, lin=True), np.fft.fftshift(P.sum(axis=0)), rtol=1e-4)
    m = O.getWaterfall(x, sz)
    assert m.dtype == np.float64 and m.shape == (sz, nb)
    assert np.allclose(m, np.fft.fftshift(P, axes=1).T, rtol=1e-4)


# ------------------------------------------------------------------ FrameSynchronisation.jl
def brute_beta(cv, n, w_min, w_max):
    """direct evaluation of FrameSynchronisation.jl:94-112 in f64"""
    cv = cv.astype(np.float64)
    S = cv.sum()
    out = np.zeros((w_max - w_min + 1, n))
    for c in range(1, n + 1):
        for w in range(w_min, w_max + 1):
            idx = [(k - 1) % n for k in range(c - w, c + w + 1)]
            blank = 2 * cv[idx[0]] + 2 * cv[idx[-1]] + 2 * cv[idx[1:-1]].sum() if w > 0 else 2 * cv[idx].sum()
            # _Sigma after step w = 2*sum_{k=c-(w_min-1)}^{c+(w_min-1)} + 2*sum of the added pairs = 2*sum_{c-w..c+w}
            blank = 2 * cv[idx].sum()
            out[w - w_min, c - 1] = ((S - blank) / (2 * (n - w)) + blank / (2 * w)) ** 2
    return out


def test_fill_beta_against_bruteforce():
    n, w_min, w_max = 60, 3, 15
    cv = (rng.random(n) * 100).astype(np.float32)
    got = O.fill_beta(cv, n, w_min, w_max)
    ref = brute_beta(cv, n, w_min, w_max)
    assert got.shape == ref.shape and np.allclose(got, ref, rtol=2e-5)


def test_sync_constants():
    s = O.SyncXY(600, 800)
    assert (s.wmin_y, s.wmax_y, s.wmin_x, s.wmax_x) == (6, 150, 40, 200)  # FrameSynchronisation.jl:36-41
    s = O.SyncXY(1235, 2592)
    assert (s.wmin_y, s.wmax_y, s.wmin_x, s.wmax_x) == (13, 308, 130, 648)


def test_fir_is_causal_zero_state():
    scipy_signal = pytest.importorskip("scipy.signal")
    s = O.SyncXY(64, 100)
    img = np.asfortranarray(rng.random((64, 100), dtype=np.float32))
    cv, ch = s.project(img)
    h = np.exp(-2 * np.arange(-2, 3) ** 2 / 25.0)
    h /= h.sum()
    assert np.allclose(cv, scipy_signal.lfilter(h, [1.0], img.astype(np.float64).sum(axis=0)), rtol=1e-5)
    assert np.allclose(ch, scipy_signal.lfilter(h, [1.0], img.astype(np.float64).sum(axis=1)), rtol=1e-5)


def test_vsync_band_image_and_stale_sy():
    h, w = 120, 200
    img = np.full((h, w), 0.2, np.float32)
    img[30:40, :] = 1.0   # bright horizontal band, centre row ~35 (1-based 35.5), +2 from the causal FIR
    img[:, 90:120] = 1.0  # bright vertical band, centre col ~105
    img = np.asfortranarray(img)
    s = O.SyncXY(h, w)
    sy0, sx0 = s.vsync(img)
    assert sy0 == 1                      # beta_y was all zeros when it was read (reference ordering :66)
    assert abs(sx0 - (105 + 2)) <= 2     # column-band centre shifted by the FIR group delay
    sy1, sx1 = s.vsync(img)
    assert sx1 == sx0 and abs(sy1 - (35 + 2)) <= 2  # s_y arrives one call late
    # argmax equals the brute-force evaluation of the same beta
    cv, ch = s.project(img)
    bx = brute_beta(cv, w, s.wmin_x, s.wmax_x)
    assert sx0 == int(np.unravel_index(np.argmax(bx.T), bx.T.shape)[0]) + 1
    s.reset()
    assert s.vsync(img)[0] == 1


def test_circshift_matches_julia_semantics():
    img = np.asfortranarray(rng.random((6, 9), dtype=np.float32))
    out = O.circshift_neg(img, 2, 4)  # circshift(image,(-2,-4)): out[i,j] = in[i+2, j+4]
    assert np.array_equal(out, np.roll(img, (-2, -4), axis=(0, 1)))


# ------------------------------------------------------------------ frame loop = composition of its parts
def test_frames_equals_composition():
    S, y_t, x_t, nfr = 700, 40, 52, 3
    iq = (rng.standard_normal(S * nfr + 13) + 1j * rng.standard_normal(S * nfr + 13)).astype(np.complex64)
    alpha = np.float32(0.1)
    st = np.zeros((600, 800), np.float32, order="F")
    out = O.frames(O.SyncXY(600, 800), iq, S, y_t, x_t, alpha, st, want_raster=True)
    assert out["n_frames"] == nfr
    sync = O.SyncXY(600, 800)
    acc = np.zeros((600, 800), np.float32)
    sig = O.amDemod(iq)
    for f in range(nfr):
        ras = O.sig_to_image(sig[f * S:(f + 1) * S], y_t, x_t)
        assert np.array_equal(ras, out["raster"][f])
        img = O.downgradeImage(ras)
        sy, sx = sync.vsync(img)
        assert (sy, sx) == tuple(out["sync_idx"][f])
        img = O.circshift_neg(img, sy, sx)
        acc = (alpha * acc + (np.float32(1) - alpha) * img).astype(np.float32)
        assert np.array_equal(acc, out["frames"][f])
    assert np.array_equal(acc, st)


# ------------------------------------------------------------------ init_resampler / initLPF
def _julia_lpf_phase(N):
    """theta[k] of Resampler.jl:88-90 as Base's range arithmetic evaluates it (see lpf_phase_step in the oracle), restated
    independently: error-free products through Fractions instead of fma."""
    import math
    import struct
    from fractions import Fraction

    def truncbits(x, nb):
        b = struct.unpack("<Q", struct.pack("<d", x))[0] & ~((1 << nb) - 1)
        return struct.unpack("<d", struct.pack("<Q", b))[0]

    def canon(big, little):
        h = big + little
        return h, (big - h) + little

    two_pi, y, g = 2 * math.pi, float(N), -(N - 1) / 2.0
    hi = two_pi / y
    uh = hi * y
    ul = float(Fraction(hi) * Fraction(y) - Fraction(uh))          # mul12's low word
    hi, lo = canon(hi, (((two_pi - uh) - ul) + 0.0) / y)
    nb = 0 if N < 2 else min(27, math.ceil(math.log2(N - 1)))
    hi_t = truncbits(hi, nb)
    lo_t = (hi - hi_t) + lo
    hi2, lo2 = canon(hi_t * g, lo_t * g)
    return np.array([k * hi2 + k * lo2 for k in range(N)])


@pytest.mark.parametrize("n,up", [(500, 4), (6, 1), (1218, 2), (2691, 1), (6561, 2), (3000, 2), (18, 1), (24, 2), (1023, 1)])
def test_init_lpf_properties(n, up):
    """(6, 1), (1218, 2), (2691, 1), (6561, 2), (3000, 2), ...: sizeFFT divisible by 6 (or 3 with upCoeff 1) --
    round.(exp(im*theta)) (Resampler.jl:90) then has an entry whose sine or cosine is 0.5 -/+ 1e-13, decided by the last bit of
    theta, i.e. by Base's TwicePrecision range arithmetic ([RECALLED], restated in _julia_lpf_phase and in the oracle)."""
    scipy_signal = pytest.importorskip("scipy.signal")
    r = O.Resampler(n, up)
    H = r.lpf()
    N = n * up
    # restate Resampler.jl:83-99 in numpy (f64) and compare
    H0 = np.zeros(N, complex)
    bound = int(np.round(N / up / 2))
    th = _julia_lpf_phase(N)
    near = np.minimum(np.abs(np.abs(np.cos(th[:bound])) - 0.5), np.abs(np.abs(np.sin(th[:bound])) - 0.5))
    if N % 6 == 0 and up <= 2 and N > 6:
        assert near.min() < 1e-11     # the case this size is here for
    H0[:bound] = np.round(np.cos(th[:bound])) + 1j * np.round(np.sin(th[:bound]))
    h = np.fft.ifft(H0) * scipy_signal.windows.blackman(N, sym=True)
    ref = np.fft.fft(h) * (-1.0) ** np.arange(N)
    assert np.max(np.abs(H - ref)) < 1e-6 * np.max(np.abs(ref))
    x = rng.standard_normal(n).astype(np.float32)
    out = np.empty(N, np.float32)
    r(out, x)
    stuffed = np.zeros(N, complex)
    stuffed[::up] = x
    ref_out = 2 * up * np.real(np.fft.ifft(np.fft.fft(stuffed) * ref))
    assert np.max(np.abs(out - ref_out)) < 1e-5 * np.max(np.abs(ref_out))
    with pytest.raises(AssertionError):
        r(out, x[:-1])
