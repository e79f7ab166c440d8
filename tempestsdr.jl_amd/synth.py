"""Deterministic synthetic HDMI/VGA leak for tests and bench.py.

A test card (bars + checker + text-like block) sits inside an x_t x y_t TOTAL raster whose
blanking is white (the reference's vsync looks for a bright band: arg_area_blank = findmax,
FrameSynchronisation.jl:53).  The pixel stream runs at f_pix = x_t*y_t*fv; each IQ sample is
the box average of the pixel stream over one sample period, AM-modulated on a slowly rotating
carrier, plus complex Gaussian noise.  Everything is keyed by a counter-based RNG
(SplitMix64 of seed and sample index) so any slice can be regenerated identically.

This is test/bench infrastructure; it is not part of the reference's API.
"""
import numpy as np

SEED = 20251017
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """SplitMix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed, idx, stream=0):
    """float64 in (0,1), a pure function of (seed, stream, idx)."""
    with np.errstate(over="ignore"):
        key = splitmix64(np.uint64(seed) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
        h = splitmix64(np.asarray(idx, dtype=np.uint64) ^ key)
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def gaussian_complex(seed, n0, n):
    """n unit-variance complex Gaussians for sample indices n0..n0+n-1 (Box-Muller in f64)."""
    idx = np.arange(n0, n0 + n, dtype=np.uint64)
    u1 = uniform01(seed, idx, 1)
    u2 = uniform01(seed, idx, 2)
    r = np.sqrt(-np.log(u1))  # variance 1/2 per component
    return r * np.cos(2 * np.pi * u2) + 1j * r * np.sin(2 * np.pi * u2)


# Blanking profiles.  The reference's vsync scores a blank-band centre c and half-width w by the mean of the window
# [c-w, c+w] (FrameSynchronisation.jl:101-108), w >= 5 % of the 800 columns / 1 % of the 600 rows.
#   "plateau": the whole blanking interval at the constant level 1.0 (SURVEY 8d as first written).  Every window that
#              fits inside the band then has the same mean, so beta is flat over dozens of centres, the winner is decided
#              by noise at the 1e-6 level and even exact f32 ties occur: a regression input for the sync guard, not a
#              workload with a defined answer.
#   "box"    : porches at 0.3 and, inside the blanking interval, one bright (1.0) band exactly as wide as the narrowest
#              window vsync tries (81 of 800 columns, 13 of 600 rows), placed off-centre.  The window sum is then a
#              triangle in c with its apex on the band -- a clear, unique answer.  (Frames of one buffer still drift by
#              Fs/fv - round(Fs/fv) samples each, so the apex crosses the column grid: C2's default offset is chosen so
#              that the ten phases of its 0.9-column drift stay 0.05 columns away from a two-column tie.)
BLANK_BOX = dict(porch=0.3, bx=0.1013, by=0.0217, px=0.35, py=0.3, row0_frac=400 / 1125, col0_frac=1101 / 2576)


def test_card(x_t, y_t, active_w=None, active_h=None, row0=0, col0=0, seed=SEED, blank="box"):
    """(y_t, x_t) float32 in [0,1]: test card in the active area, blanking per `blank` ("box" or "plateau", above), then
    rolled by (row0, col0) so start-of-frame sits at a known non-zero offset."""
    if active_w is None:
        active_w = int(round(x_t * 0.78))
    if active_h is None:
        active_h = int(round(y_t * 0.955))
    img = np.ones((y_t, x_t), np.float32)
    if blank == "box":
        b = BLANK_BOX
        nbx, nby = x_t - active_w, y_t - active_h
        wx, wy = min(nbx, int(round(b["bx"] * x_t))), min(nby, int(round(b["by"] * y_t)))
        tx = np.full(nbx, b["porch"], np.float32)
        sx = int(round(b["px"] * (nbx - wx)))
        tx[sx:sx + wx] = 1.0
        ty = np.full(nby, b["porch"], np.float32)
        sy = int(round(b["py"] * (nby - wy)))
        ty[sy:sy + wy] = 1.0
        img[:, active_w:] = tx[None, :]
        img[active_h:, :] = ty[:, None]
        img[active_h:, active_w:] = np.maximum(ty[:, None], tx[None, :])
    elif blank != "plateau":
        raise ValueError(f"unknown blanking profile {blank!r}")
    yy, xx = np.mgrid[0:active_h, 0:active_w]
    bars = (7 - (xx * 8) // active_w).astype(np.float32) / 7.0 * 0.8  # 8 grey bars, bright -> dark
    card = bars.copy()
    # checkerboard block in the lower-left third
    cb = max(2, active_w // 96)
    chk = (((xx // cb) + (yy // cb)) & 1).astype(np.float32) * 0.7
    m = (yy > active_h * 0.62) & (xx < active_w * 0.45)
    card[m] = chk[m]
    # "text-like" block: pseudo-random on/off runs, high horizontal frequency
    txt = (uniform01(seed, (yy // max(2, active_h // 90)) * 7919 + xx // 3, 7) > 0.55).astype(np.float32) * 0.75
    m = (yy > active_h * 0.15) & (yy < active_h * 0.5) & (xx > active_w * 0.55) & (xx < active_w * 0.95)
    card[m] = txt[m]
    img[:active_h, :active_w] = card
    return np.roll(img, (row0, col0), axis=(0, 1))


def synth_leak(Fs, x_t, y_t, fv, n_samples, *, n0=0, card="box", seed=SEED, snr_db=20.0, amp=5e-3, df=1e3,
               phi0=0.3, row0=None, col0=None, chunk=1 << 20):
    """complex64 IQ[n0 : n0+n_samples] of the synthetic leak.  card: a blanking profile name ("box", the default, or
    "plateau": see test_card) or a ready (y_t, x_t) image.  row0 / col0: start-of-frame offset in raster lines / pixels
    (defaults: 37 / 211 for "plateau", the fractions of BLANK_BOX for "box")."""
    if card is None:
        card = "box"
    if isinstance(card, str):
        if row0 is None:
            row0 = 37 if card == "plateau" else int(round(BLANK_BOX["row0_frac"] * y_t))
        if col0 is None:
            col0 = 211 if card == "plateau" else int(round(BLANK_BOX["col0_frac"] * x_t))
        card = test_card(x_t, y_t, row0=row0 % y_t, col0=col0 % x_t, seed=seed, blank=card)
    flat = card.reshape(-1).astype(np.float64)  # line-major pixel stream of one frame
    P = flat.size
    cs = np.concatenate(([0.0], np.cumsum(flat)))
    total = cs[-1]
    r = P * fv / Fs  # pixels per sample
    sig_pow = None
    out = np.empty(n_samples, np.complex64)

    def cs_abs(u):
        k = np.floor(u / P)
        rem = u - k * P
        q = np.minimum(np.floor(rem).astype(np.int64), P - 1)
        return k * total + cs[q] + (rem - q) * flat[q]

    for a in range(0, n_samples, chunk):
        b = min(a + chunk, n_samples)
        n = np.arange(n0 + a, n0 + b, dtype=np.float64)
        v = (cs_abs((n + 1.0) * r) - cs_abs(n * r)) / r
        env = 0.1 + 0.9 * v
        ph = phi0 + 2 * np.pi * df * n / Fs
        s = env * (np.cos(ph) + 1j * np.sin(ph))
        if sig_pow is None:
            sig_pow = float(np.mean(np.abs(s) ** 2))
        sigma = np.sqrt(sig_pow / (10.0 ** (snr_db / 10.0)))
        s = s + sigma * gaussian_complex(seed, n0 + a, b - a)
        out[a:b] = (amp * s).astype(np.complex64)
    return out


# named workloads
WORKLOADS = {
    "C2": dict(Fs=20e6, x_t=2576, y_t=1125, fv=60.0, acquisition=0.5),   # 1080p60 @ 20 MS/s
    "C3": dict(Fs=200e6, x_t=2576, y_t=1125, fv=60.0, acquisition=0.5),  # 10x oversampled
    "C5": dict(Fs=50e6, x_t=4400, y_t=2250, fv=60.0, acquisition=0.5),   # 4K60 (not in the reference table)
    # experiment only: column height a multiple of 32 floats, so every 256-byte store segment is line-aligned
    "C2A": dict(Fs=20e6, x_t=2576, y_t=1152, fv=60.0, acquisition=0.5),
}


def samples_per_frame(Fs, fv):
    """image_size_down = round(Fs/fv)  (GUI.jl:103-109)"""
    return int(np.round(Fs / fv))
