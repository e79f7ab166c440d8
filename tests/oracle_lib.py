"""ctypes binding of oracle/libtempest_oracle.so -- the CPU restatement of the reference.

TEST INFRASTRUCTURE.  Imported only by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
OLIB = os.path.join(ODIR, "libtempest_oracle.so")

_lib = None
vp, sz, ci, cd, cf = C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_float


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ODIR, "tempest_oracle.c")
        if not os.path.exists(OLIB) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(OLIB)):
            subprocess.run(["make", "-C", ODIR], check=True, capture_output=True)
        L = C.CDLL(OLIB)
        L.orc_sync_create.restype = vp
        L.orc_sync_create.argtypes = [ci, ci]
        L.orc_sync_beta_x.restype = vp
        L.orc_sync_beta_y.restype = vp
        L.orc_resampler_init.restype = vp
        L.orc_resampler_init.argtypes = [sz, ci]
        L.orc_resampler_H.restype = vp
        for n in ("orc_sync_reset", "orc_sync_free", "orc_sync_bounds", "orc_sync_beta_x", "orc_sync_beta_y",
                  "orc_resampler_free", "orc_resampler_H"):
            getattr(L, n).argtypes = [vp] + ([vp] if n == "orc_sync_bounds" else [])
        _lib = L
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _chk(rc, what):
    if rc == -1:
        raise AssertionError(what)
    if rc == -2:
        raise IndexError(what)
    if rc != 0:
        raise RuntimeError(f"{what}: rc={rc}")


def _c64(z):
    return np.ascontiguousarray(z, dtype=np.complex64)


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def amDemod(sig):
    z = _c64(sig); out = np.empty(z.size, np.float32)
    lib().orc_am_demod(_p(z), sz(z.size), _p(out)); return out


def abs2(sig):
    z = _c64(sig); out = np.empty(z.size, np.float32)
    lib().orc_abs2(_p(z), sz(z.size), _p(out)); return out


def invert_amDemod(sig):
    z = _c64(sig); out = np.empty(z.size, np.float32)
    _chk(lib().orc_invert_am(_p(z), sz(z.size), _p(out)), "invert_amDemod"); return out


def fmDemod(sig):
    z = _c64(sig); out = np.empty(z.size, np.float32)
    lib().orc_fm_demod(_p(z), sz(z.size), _p(out)); return out


def imresize1d(sig, n_out):
    x = _f32(sig); out = np.empty(int(n_out), np.float32)
    _chk(lib().orc_resize1d(_p(x), sz(x.size), sz(int(n_out)), _p(out)), "imresize"); return out


def sig_to_image(sig, y_t, x_t):
    x = _f32(sig); img = np.empty((y_t, x_t), np.float32, order="F")
    _chk(lib().orc_sig_to_image(_p(x), sz(x.size), ci(y_t), ci(x_t), _p(img)), "sig_to_image"); return img


def imresize2d(image, size):
    a = np.asfortranarray(image, dtype=np.float32)
    out = np.empty((size[0], size[1]), np.float32, order="F")
    _chk(lib().orc_resize2d(_p(a), ci(a.shape[0]), ci(a.shape[1]), ci(size[0]), ci(size[1]), _p(out)), "imresize2d")
    return out


def downgradeImage(image):
    return imresize2d(image, (600, 800))


def naiveResampler(sigId, up):
    x = _f32(sigId); out = np.empty(x.size * up, np.float32)
    lib().orc_naive_resample(_p(x), sz(x.size), ci(up), _p(out)); return out


def fft(x, inverse=False):
    a = np.ascontiguousarray(x, dtype=np.complex128).copy()
    _chk(lib().orc_fft_c64(_p(a), sz(a.size), ci(1 if inverse else -1)), "fft"); return a


def calculate_autocorrelation(x, Fs, minDelay, maxDelay, scale="log"):
    xv = _f32(x)
    imin = 1 + int(np.round(minDelay * Fs)); imax = int(np.round(maxDelay * Fs))
    out = np.empty(max(imax - imin + 1, 1), np.float32); n = sz(0)
    _chk(lib().orc_autocorr(_p(xv), sz(xv.size), cd(Fs), cd(minDelay), cd(maxDelay), ci(1 if scale == "log" else 0),
                            _p(out), C.byref(n)), "calculate_autocorrelation")
    lags = np.arange(0, imax - imin + 1, dtype=np.float64) / Fs
    return out[: n.value], lags


def zoom_bounds(N, Fs, rate_min, rate_max):
    a, b = sz(0), sz(0)
    _chk(lib().orc_zoom_bounds(sz(N), cd(Fs), cd(rate_min), cd(rate_max), C.byref(a), C.byref(b)), "zoom_autocorr")
    return a.value, b.value


def zoom_autocorr(G, Fs, rate_min=20, rate_max=100):
    a, b = zoom_bounds(len(G), Fs, rate_min, rate_max)
    idx = np.arange(a, b + 1, dtype=np.float64)
    return 1.0 / (idx / Fs), np.asarray(G)[a - 1: b]


def _sig(sig):
    a = np.ascontiguousarray(sig)
    if np.iscomplexobj(a):
        return a.astype(np.complex64), 1
    return a.astype(np.float32), 0


def getSpectrum(sig, N=None, lin=False):
    a, c = _sig(sig); N = a.size if N is None else N
    y = np.empty(N, np.float32)
    _chk(lib().orc_spectrum(_p(a), ci(c), sz(N), ci(int(lin)), _p(y)), "getSpectrum"); return y


def getWelch(sig, sizeFFT=1024, lin=False):
    a, c = _sig(sig); y = np.empty(sizeFFT, np.float32)
    _chk(lib().orc_welch(_p(a), ci(c), sz(a.size), sz(sizeFFT), ci(int(lin)), _p(y)), "getWelch"); return y


def getWaterfall(sig, sizeFFT=1024):
    a, c = _sig(sig); nb = a.size // sizeFFT
    m = np.empty((sizeFFT, nb), np.float64, order="F")
    _chk(lib().orc_waterfall(_p(a), ci(c), sz(a.size), sz(sizeFFT), _p(m)), "getWaterfall"); return m


class SyncXY:
    def __init__(self, y_t, x_t):
        self.h = lib().orc_sync_create(y_t, x_t)
        if not self.h:
            raise AssertionError("orc_sync_create")
        self.y_t, self.x_t = y_t, x_t
        b = (C.c_int * 4)(); lib().orc_sync_bounds(self.h, b)
        self.wmin_y, self.wmax_y, self.wmin_x, self.wmax_x = list(b)

    def vsync(self, image):
        a = np.asfortranarray(image, dtype=np.float32); sy, sx = ci(0), ci(0)
        _chk(lib().orc_vsync(vp(self.h), _p(a), C.byref(sy), C.byref(sx)), "vsync"); return sy.value, sx.value

    def project(self, image):
        a = np.asfortranarray(image, dtype=np.float32)
        cv = np.empty(self.x_t, np.float32); ch = np.empty(self.y_t, np.float32)
        _chk(lib().orc_project(vp(self.h), _p(a), _p(cv), _p(ch)), "project"); return cv, ch

    def beta(self, which):
        if which == "x":
            shape, ptr = (1 + self.wmax_x - self.wmin_x, self.x_t), lib().orc_sync_beta_x(self.h)
        else:
            shape, ptr = (1 + self.wmax_y - self.wmin_y, self.y_t), lib().orc_sync_beta_y(self.h)
        n = shape[0] * shape[1]
        flat = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,)).copy()
        return flat.reshape(shape, order="F")

    def reset(self):
        lib().orc_sync_reset(self.h)

    def __del__(self):
        try:
            lib().orc_sync_free(self.h)
        except Exception:
            pass


def fill_beta(cv, n, w_min, w_max):
    x = _f32(cv); beta = np.empty((w_max - w_min + 1, n), np.float32, order="F")
    lib().orc_fill_beta(_p(beta), _p(x), ci(n), ci(w_min), ci(w_max)); return beta


def circshift_neg(image, s_y, s_x):
    a = np.asfortranarray(image, dtype=np.float32); out = np.empty_like(a, order="F")
    lib().orc_circshift_neg(_p(a), ci(a.shape[0]), ci(a.shape[1]), ci(s_y), ci(s_x), _p(out)); return out


def frames(sync, iq, S, y_t, x_t, alpha, imageOut, do_align=True, want_frames=True, want_raster=False):
    z = _c64(iq); nb = z.size // S
    fr = np.empty((nb, 480000), np.float32) if want_frames else None
    ra = np.empty((nb, y_t * x_t), np.float32) if want_raster else None
    idx = np.zeros((nb, 2), np.int32); n = ci(0)
    _chk(lib().orc_frames(vp(sync.h if sync is not None else 0), _p(z), sz(z.size), sz(S), ci(y_t), ci(x_t), cf(alpha),
                          ci(int(do_align)), _p(imageOut), _p(fr), _p(ra), _p(idx), C.byref(n)), "frames")
    out = {"n_frames": n.value, "sync_idx": idx}
    if fr is not None:
        out["frames"] = [fr[f].reshape((600, 800), order="F") for f in range(nb)]
    if ra is not None:
        out["raster"] = [ra[f].reshape((y_t, x_t), order="F") for f in range(nb)]
    return out


class Resampler:
    def __init__(self, bufferSize, up):
        self.h = lib().orc_resampler_init(bufferSize, up)
        if not self.h:
            raise AssertionError("orc_resampler_init")
        self.n, self.up = bufferSize, up

    def __call__(self, out, inp):
        x = _f32(inp)
        _chk(lib().orc_resampler_run(vp(self.h), _p(x), sz(x.size), _p(out)), "resampler!")

    def lpf(self):
        ptr = lib().orc_resampler_H(self.h)
        flat = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(2 * self.n * self.up,)).copy()
        return flat[0::2] + 1j * flat[1::2]

    def __del__(self):
        try:
            lib().orc_resampler_free(self.h)
        except Exception:
            pass
