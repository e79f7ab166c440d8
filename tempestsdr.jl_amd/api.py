"""Host-side mirror of TempestSDR.jl's processing API over the HIP C ABI.

Same function names, argument meaning and error behaviour as the reference's
Demodulation.jl / Resampler.jl / Autocorrelations.jl / GetSpectrum.jl /
FrameSynchronisation.jl, so tests read like the reference's own call sites
(GUI.jl:73-74,136,164,168,171; production/investigate_data.jl).  Matrices are Fortran-order
numpy arrays (Julia is column-major); complex vectors are complex64 (ComplexF32).

Every function runs on the GPU through libtempest_hip.so; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RENDER_H, RENDER_W, TempestHIPError, check


def _ptr(a):
    """host numpy array, torch tensor (device or host) or raw integer address -> c_void_p"""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"cannot take the address of {type(a)}")


def _c64(sig):
    a = np.ascontiguousarray(sig)
    if not np.iscomplexobj(a):
        # amDemod(sig::Array{Complex{T}}): a real array is a MethodError in the reference
        raise AssertionError("expected a complex vector (MethodError in the reference)")
    return a.astype(np.complex64, copy=False)


def _f32(sig):
    return np.ascontiguousarray(sig, dtype=np.float32)


class Context:
    """One tsdr_ctx (HIP stream + workspaces).  One per caller thread, as in the reference's
    two-task layout."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        self.h = self.lib.tsdr_create(int(device))
        if not self.h:
            raise TempestHIPError(f"tsdr_create({device}) failed: no usable HIP device (no CPU fallback exists)")
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsdr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -------------------------------------------------------------------
    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.h, *args)
        check(self.h, rc, name)

    def synchronize(self):
        self.call("tsdr_synchronize")

    def set_precision(self, mode):
        """'exact' (bit-identical to the oracle) or 'fast' (within 1 ulp, default) -- tsdr_set_precision"""
        self.call("tsdr_set_precision", {"exact": _lib.EXACT, "fast": _lib.FAST}[mode])

    @property
    def precision(self):
        return "exact" if self.lib.tsdr_get_precision(self.h) == _lib.EXACT else "fast"

    def set_option(self, name, value):
        """switches ("ac_mixed", "fft_no_mix2", "sync_guard_ppb", "sync_guard_auto", ...) -- tsdr_set_option"""
        self.call("tsdr_set_option", name.encode(), int(value))

    def sync_guard_stats(self, reset=False):
        """running totals of the FAST frame loop's sync guard: (frames checked, frames flagged = computed exactly)"""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self.call("tsdr_sync_guard_stats", C.byref(a), C.byref(b), int(bool(reset)))
        return int(a.value), int(b.value)

    def sync_guard_auto(self):
        """adaptive route of the sync guard: (whole buffers run exactly right now?, calls that did, route changes)"""
        e, a, b = C.c_int(0), C.c_ulonglong(0), C.c_ulonglong(0)
        self.call("tsdr_sync_guard_auto", C.byref(e), C.byref(a), C.byref(b))
        return bool(e.value), int(a.value), int(b.value)

    def wait_stats(self):
        """(stream waits given up after "wait_ms", guard ring entries that went uncounted) on this context -- tsdr_wait_stats"""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self.call("tsdr_wait_stats", C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def pipeline_info(self):
        """what tsdr_frames_submit_d measured on this context: dict(trials_left, chosen, ms_per_buffer[8], text)"""
        left, chosen, ms, text = C.c_int(0), C.c_int(-1), (C.c_float * 8)(), C.create_string_buffer(1024)
        self.call("tsdr_frames_pipeline_info", C.byref(left), C.byref(chosen), ms, 8, text, 1024)
        txt = text.value.decode()
        import re
        m = re.search(r"measurements started on this context: (\d+)", txt)
        return {"trials_left": left.value, "chosen": chosen.value, "ms_per_buffer": [round(float(v), 5) for v in ms],
                "text": txt, "measurements_started": int(m.group(1)) if m else None}

    def sync_guard_margins(self, max_frames=1 << 16):
        """(frames, 2) relative top-2 margins (x, y) the guard saw in the last FAST frame-loop call"""
        n = C.c_int(0)
        self.call("tsdr_sync_guard_margins", 0, None, C.byref(n))
        nf = min(n.value, int(max_frames))
        m = np.zeros((nf, 2), np.float32)
        if nf:
            self.call("tsdr_sync_guard_margins", nf, _ptr(m), C.byref(n))
        return m

    def set_stream(self, stream_ptr):
        self.call("tsdr_set_stream", C.c_void_p(stream_ptr or 0))

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int(0)
        mem = C.c_size_t(0)
        self.call("tsdr_device_info", name, 256, C.byref(cu), C.byref(mem))
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}

    # -- resident device buffers (for the *_d entry points) ----------------------------
    def dev_alloc(self, nbytes):
        p = self.lib.tsdr_dev_alloc(self.h, int(nbytes))
        if not p:
            raise MemoryError(self.lib.tsdr_last_error(self.h).decode())
        return p

    def dev_free(self, p):
        self.call("tsdr_dev_free", C.c_void_p(p))

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        p = self.dev_alloc(a.nbytes)
        self.call("tsdr_upload", C.c_void_p(p), _ptr(a), a.nbytes)
        return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self.call("tsdr_download", _ptr(out), C.c_void_p(p), out.nbytes)
        return out

    def timer_start(self):
        self.call("tsdr_timer_start")

    def timer_stop(self):
        ms = C.c_double(0)
        self.call("tsdr_timer_stop", C.byref(ms))
        return ms.value

    def profile(self, on):
        self.call("tsdr_profile_enable", int(bool(on)))

    def profile_reset(self):
        self.call("tsdr_profile_reset")

    def profile_results(self):
        n = self.lib.tsdr_profile_count(self.h)
        if n < 0:
            check(self.h, n, "tsdr_profile_count")
        out = {}
        for i in range(n):
            name = C.create_string_buffer(128)
            ms = C.c_double(0)
            cnt = C.c_longlong(0)
            self.call("tsdr_profile_get", i, name, 128, C.byref(ms), C.byref(cnt))
            out[name.value.decode()] = {"total_ms": ms.value, "launches": cnt.value}
        return out

    # -- Demodulation.jl ----------------------------------------------------------------
    def amDemod(self, sig):
        z = _c64(sig)
        out = np.empty(z.shape, np.float32)
        self.call("tsdr_am_demod", _ptr(z), z.size, _ptr(out))
        return out

    def invert_amDemod(self, sig):
        z = _c64(sig)
        out = np.empty(z.shape, np.float32)
        self.call("tsdr_invert_am", _ptr(z), z.size, _ptr(out))
        return out

    def fmDemod(self, sig):
        z = _c64(sig)
        out = np.empty(z.shape, np.float32)
        self.call("tsdr_fm_demod", _ptr(z), z.size, _ptr(out))
        return out

    def abs2(self, sig):
        z = _c64(sig)
        out = np.empty(z.shape, np.float32)
        self.call("tsdr_abs2", _ptr(z), z.size, _ptr(out))
        return out

    # -- Resampler.jl -------------------------------------------------------------------
    def imresize1d(self, sig, n_out):
        x = _f32(sig)
        out = np.empty(int(n_out), np.float32)
        self.call("tsdr_resize1d", _ptr(x), x.size, int(n_out), _ptr(out))
        return out

    def sig_to_image(self, sig, y_t, x_t):
        x = _f32(sig)
        img = np.empty((int(y_t), int(x_t)), np.float32, order="F")
        self.call("tsdr_sig_to_image", _ptr(x), x.size, int(y_t), int(x_t), _ptr(img))
        return img

    def imresize2d(self, image, size):
        a = np.asfortranarray(image, dtype=np.float32)
        h, w = int(size[0]), int(size[1])
        out = np.empty((h, w), np.float32, order="F")
        self.call("tsdr_resize2d", _ptr(a), a.shape[0], a.shape[1], h, w, _ptr(out))
        return out

    def downgradeImage(self, image):
        return self.imresize2d(image, (RENDER_H, RENDER_W))

    def naiveResampler(self, sigOut, sigId, upCoeff):
        x = _f32(sigId)
        if not (isinstance(sigOut, np.ndarray) and sigOut.dtype == np.float32 and sigOut.flags.c_contiguous):
            raise AssertionError("sigOut must be a contiguous float32 array")
        if sigOut.size < x.size * int(upCoeff):
            raise IndexError("sigOut too short (BoundsError in the reference)")
        self.call("tsdr_naive_resample", _ptr(x), x.size, int(upCoeff), _ptr(sigOut))

    def init_resampler(self, T, bufferSize, upCoeff):
        """init_resampler(T,bufferSize,upCoeff) -> resampler!(out,in)  (Resampler.jl:26-62)"""
        if np.dtype(T) != np.float32:
            raise AssertionError("only Float32 resamplers are implemented on the GPU path")
        return Resampler(self, int(bufferSize), int(upCoeff))

    # -- Autocorrelations.jl --------------------------------------------------------------
    def calculate_autocorrelation(self, x, Fs, minDelay, maxDelay, scale="log"):
        xv = _f32(x)
        index_min = 1 + int(np.round(minDelay * Fs))
        index_max = int(np.round(maxDelay * Fs))
        cnt = max(index_max - index_min + 1, 0)
        out = np.empty(max(cnt, 1), np.float32)
        n_out = C.c_size_t(0)
        self.call("tsdr_autocorr", _ptr(xv), xv.size, float(Fs), float(minDelay), float(maxDelay),
                  1 if scale == "log" else 0, _ptr(out), C.byref(n_out))
        lags = np.arange(0, index_max - index_min + 1, dtype=np.float64) * (1.0 / Fs)
        return out[: n_out.value], lags

    def autocorr_search(self, sig, Fs, minDelay, maxDelay, rate_min=50, rate_max=90, scale="log"):
        """calculate_autocorrelation + zoom_autocorr + findmax as ONE library call (tsdr_autocorr_search_d; GUI.jl:73-81).
        sig: real power samples, or complex IQ whose abs2 is formed on the fly (GUI.jl:70).
        -> (G, pos, val): the lag vector, the 0-based findmax position inside the zoom window, its value."""
        a = np.ascontiguousarray(sig)
        is_iq = int(np.iscomplexobj(a))
        a = a.astype(np.complex64 if is_iq else np.float32, copy=False)
        index_min = 1 + int(np.round(minDelay * Fs))
        index_max = int(np.round(maxDelay * Fs))
        cnt = max(index_max - index_min + 1, 0)
        pmin, pmax = C.c_size_t(0), C.c_size_t(0)
        check(self.h, self.lib.tsdr_zoom_bounds(cnt, float(Fs), float(rate_min), float(rate_max), C.byref(pmin), C.byref(pmax)),
              "tsdr_zoom_bounds")
        d_in, d_out = self.upload(a), self.dev_alloc(max(cnt, 1) * 4)
        try:
            n_out, idx, val = C.c_size_t(0), C.c_size_t(0), C.c_float(0)
            self.call("tsdr_autocorr_search_d", C.c_void_p(d_in), is_iq, a.size, float(Fs), float(minDelay), float(maxDelay),
                      1 if scale == "log" else 0, C.c_void_p(d_out), C.byref(n_out), int(pmin.value - 1),
                      int(pmax.value - pmin.value + 1), C.byref(idx), C.byref(val))
            G = self.download(d_out, (n_out.value,), np.float32)
        finally:
            self.dev_free(d_in)
            self.dev_free(d_out)
        return G, int(idx.value), float(val.value)

    def zoom_autocorr(self, G, Fs, rate_min=20, rate_max=100):
        pmin, pmax = C.c_size_t(0), C.c_size_t(0)
        rc = self.lib.tsdr_zoom_bounds(len(G), float(Fs), float(rate_min), float(rate_max), C.byref(pmin), C.byref(pmax))
        check(self.h, rc, "tsdr_zoom_bounds")
        idx = np.arange(pmin.value, pmax.value + 1, dtype=np.float64)
        rates = 1.0 / (idx / Fs)
        return rates, np.asarray(G)[pmin.value - 1: pmax.value]

    # -- GetSpectrum.jl -----------------------------------------------------------------
    def _sig(self, sig):
        a = np.ascontiguousarray(sig)
        if np.iscomplexobj(a):
            return a.astype(np.complex64, copy=False), 1
        return a.astype(np.float32, copy=False), 0

    def getSpectrum(self, fs, sig, N=None, lin=False):
        a, cplx = self._sig(sig)
        N = a.size if N is None else int(N)
        if N > a.size:
            raise IndexError("N exceeds the signal length (BoundsError in the reference)")
        y = np.empty(N, np.float32)
        self.call("tsdr_spectrum", _ptr(a), cplx, N, int(lin), _ptr(y))
        freq = (np.arange(N) / N - 0.5) * fs
        return freq, y

    def getWelch(self, fe, sig, sizeFFT=1024, lin=False):
        a, cplx = self._sig(sig)
        y = np.empty(int(sizeFFT), np.float32)
        self.call("tsdr_welch", _ptr(a), cplx, a.size, int(sizeFFT), int(lin), _ptr(y))
        freq = (np.arange(sizeFFT) / sizeFFT - 0.5) * fe
        return freq, y

    def getWaterfall(self, fe, sig, sizeFFT=1024):
        a, cplx = self._sig(sig)
        nb = a.size // int(sizeFFT)
        m = np.empty((int(sizeFFT), nb), np.float64, order="F")
        self.call("tsdr_waterfall", _ptr(a), cplx, a.size, int(sizeFFT), _ptr(m))
        f_ax = (np.arange(sizeFFT) / sizeFFT - 0.5) * fe
        t_ax = np.arange(nb) * (sizeFFT / fe)
        return t_ax, f_ax, m

    def fft(self, x, inverse=False):
        a = np.ascontiguousarray(x).astype(np.complex64)
        batch = 1 if a.ndim == 1 else a.shape[0]
        n = a.shape[-1]
        out = np.empty_like(a)
        self.call("tsdr_fft_c2c", _ptr(a), _ptr(out), n, batch, 1 if inverse else -1)
        return out

    def fft64(self, x, inverse=False):
        """complex f64 FFT (tsdr_fft_z2z): the transform initLPF's ComplexF64 filter is built with"""
        a = np.ascontiguousarray(x).astype(np.complex128)
        out = np.empty_like(a)
        self.call("tsdr_fft_z2z", _ptr(a), _ptr(out), a.size, 1 if inverse else -1)
        return out

    # -- FrameSynchronisation.jl ------------------------------------------------------------
    def SyncXY(self, image):
        a = np.asarray(image)
        return SyncXY(self, a.shape[0], a.shape[1])

    def vsync(self, image, sync):
        return sync.vsync(image)

    def fill_beta(self, cv, n, w_min, w_max):
        x = _f32(cv)
        beta = np.empty((w_max - w_min + 1, n), np.float32, order="F")
        self.call("tsdr_fill_beta", _ptr(x), int(n), int(w_min), int(w_max), _ptr(beta))
        return beta

    def circshift_neg(self, image, s_y, s_x):
        a = np.asfortranarray(image, dtype=np.float32)
        out = np.empty_like(a, order="F")
        self.call("tsdr_circshift_neg", _ptr(a), a.shape[0], a.shape[1], int(s_y), int(s_x), _ptr(out))
        return out

    # -- frame loop (GUI.jl:163-178) ---------------------------------------------------------
    def frames(self, sync, iq, S, y_t, x_t, alpha, imageOut, do_align=True, want_frames=True, want_raster=False):
        """One SDR buffer through the steady-state loop.  imageOut (600x800 F-order float32) is
        updated in place.  Returns dict(n_frames, frames, raster, sync_idx)."""
        z = _c64(iq)
        nb = z.size // int(S)
        if not (isinstance(imageOut, np.ndarray) and imageOut.dtype == np.float32 and imageOut.flags.f_contiguous
                and imageOut.shape == (RENDER_H, RENDER_W)):
            raise AssertionError("imageOut must be a Fortran-order float32 (600,800) array")
        frames = np.empty((nb, RENDER_H, RENDER_W), np.float32) if want_frames else None
        raster = np.empty((nb, int(y_t) * int(x_t)), np.float32) if want_raster else None
        idx = np.zeros((nb, 2), np.int32)
        n = C.c_int(0)
        self.call("tsdr_frames", C.c_void_p(sync.h if sync is not None else 0), _ptr(z), z.size, int(S), int(y_t),
                  int(x_t), C.c_float(alpha), int(bool(do_align)), _ptr(imageOut), _ptr(frames), _ptr(raster), _ptr(idx),
                  C.byref(n))
        out = {"n_frames": n.value, "sync_idx": idx}
        if frames is not None:  # each frame is stored column-major (600,800)
            out["frames"] = [frames[f].reshape(-1).reshape((RENDER_H, RENDER_W), order="F") for f in range(nb)]
        if raster is not None:
            out["raster"] = [raster[f].reshape((int(y_t), int(x_t)), order="F") for f in range(nb)]
        return out


def frames_d(ctx, sync, iq, nEch, S, y_t, x_t, alpha, do_align, state, frames_out=None, raster_out=None, sync_idx=None):
    """Device-pointer form of Context.frames: every array argument is a device buffer (torch tensor
    or raw address); enqueues on the context's stream and returns without synchronising."""
    n = C.c_int(0)
    ctx.call("tsdr_frames_d", C.c_void_p(sync.h if sync is not None else 0), _ptr(iq), int(nEch), int(S), int(y_t),
             int(x_t), C.c_float(alpha), int(bool(do_align)), _ptr(state), _ptr(frames_out), _ptr(raster_out),
             _ptr(sync_idx), C.byref(n))
    return n.value


def frames_submit_d(ctx, sync, iq, nEch, S, y_t, x_t, alpha, do_align, state, frames_out=None, raster_out=None,
                    sync_idx=None):
    """frames_d pipelined across successive buffers on the library's internal streams: only enqueues; the tail of a
    buffer (statistics, guard, shift + IIR) runs beside the image launch of the next.  Up to three submissions in flight, each
    with its own outputs.  Outputs are complete after frames_flush(ctx) in stream order / ctx.synchronize() on the host."""
    n = C.c_int(0)
    ctx.call("tsdr_frames_submit_d", C.c_void_p(sync.h if sync is not None else 0), _ptr(iq), int(nEch), int(S), int(y_t),
             int(x_t), C.c_float(alpha), int(bool(do_align)), _ptr(state), _ptr(frames_out), _ptr(raster_out),
             _ptr(sync_idx), C.byref(n))
    return n.value


def frames_sc16_d(ctx, sync, iq, scale, nEch, S, y_t, x_t, alpha, do_align, state, frames_out=None, raster_out=None, sync_idx=None,
                  submit=False):
    """frames_d / frames_submit_d on a device buffer of nEch interleaved int16 (re, im) pairs: every sample is
    ComplexF32(re, im) * scale, formed in the kernels' loaders (tsdr_frames_sc16_d / tsdr_frames_submit_sc16_d)."""
    n = C.c_int(0)
    ctx.call("tsdr_frames_submit_sc16_d" if submit else "tsdr_frames_sc16_d", C.c_void_p(sync.h if sync is not None else 0), _ptr(iq),
             C.c_float(scale), int(nEch), int(S), int(y_t), int(x_t), C.c_float(alpha), int(bool(do_align)), _ptr(state),
             _ptr(frames_out), _ptr(raster_out), _ptr(sync_idx), C.byref(n))
    return n.value


def frames_flush(ctx):
    """Order the context's stream after every buffer submitted with frames_submit_d."""
    ctx.call("tsdr_frames_flush")


class Group:
    """One process, several GPUs (tsdr_group_*): one context per device and one RCCL communicator per device inside the
    library; host arrays in and out.  The Python twin of TempestHIP.jl's `HipGroup` -- what a single-process runtime such as
    the reference's (GUI.jl:380-382) holds to use every MI355X of a node.  A group of one device is valid."""

    ROUTES = {"auto": 0, "sharded": 1, "root": 2}

    def __init__(self, devices=(0,)):
        self.lib = _lib.load()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p(0)
        rc = self.lib.tsdr_group_create(devs, len(devices), C.byref(h))
        if rc or not h.value:
            raise TempestHIPError(f"tsdr_group_create({list(devices)}) failed: {self.lib.tsdr_strerror(rc).decode()} "
                                  "(no usable HIP device / RCCL communicator; there is no CPU fallback)")
        self.h = h.value
        self.devices = tuple(int(d) for d in devices)

    def _chk(self, rc, what):
        if rc == _lib.TSDR_OK:
            return
        detail = self.lib.tsdr_group_last_error(self.h).decode()
        msg = f"{what}: {self.lib.tsdr_strerror(rc).decode()}" + (f" [{detail}]" if detail else "")
        if rc == _lib.TSDR_EINVAL:
            raise AssertionError(msg)
        if rc == _lib.TSDR_EBOUNDS:
            raise IndexError(msg)
        if rc == _lib.TSDR_ENOMEM:
            raise MemoryError(msg)
        raise TempestHIPError(msg)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsdr_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.lib.tsdr_group_size(self.h)

    def set_precision(self, mode):
        self._chk(self.lib.tsdr_group_set_precision(self.h, {"exact": _lib.EXACT, "fast": _lib.FAST}[mode]), "set_precision")

    def set_option(self, name, value):
        self._chk(self.lib.tsdr_group_set_option(self.h, name.encode(), int(value)), f"set_option({name})")

    def sync_reset(self):
        self._chk(self.lib.tsdr_group_sync_reset(self.h), "sync_reset")

    def timing(self):
        """(route, [ms per-member stage incl. upload, ms collective, ms root's final stage]) of the last call"""
        r, ms = C.c_int(0), (C.c_double * 3)()
        self._chk(self.lib.tsdr_group_timing(self.h, C.byref(r), ms), "timing")
        return {1: "sharded", 2: "root"}.get(r.value, "none"), [float(v) for v in ms]

    def autocorr_search(self, sig, Fs, minDelay, maxDelay, rate_min=50, rate_max=90, scale="log", route="auto"):
        """Context.autocorr_search over the group (tsdr_group_search; GUI.jl:73-81): -> (G, pos, val)."""
        a = np.ascontiguousarray(sig)
        is_iq = int(np.iscomplexobj(a))
        a = a.astype(np.complex64 if is_iq else np.float32, copy=False)
        index_min = 1 + int(np.round(minDelay * Fs))
        index_max = int(np.round(maxDelay * Fs))
        cnt = max(index_max - index_min + 1, 0)
        pmin, pmax = C.c_size_t(0), C.c_size_t(0)
        check(None, self.lib.tsdr_zoom_bounds(cnt, float(Fs), float(rate_min), float(rate_max), C.byref(pmin), C.byref(pmax)),
              "tsdr_zoom_bounds")
        G = np.empty(max(cnt, 1), np.float32)
        n_out, idx, val = C.c_size_t(0), C.c_size_t(0), C.c_float(0)
        self._chk(self.lib.tsdr_group_search(self.h, _ptr(a), is_iq, a.size, float(Fs), float(minDelay), float(maxDelay),
                                             1 if scale == "log" else 0, _ptr(G), C.byref(n_out), int(pmin.value - 1),
                                             int(pmax.value - pmin.value + 1), C.byref(idx), C.byref(val), self.ROUTES[route]),
                  "tsdr_group_search")
        return G[: n_out.value], int(idx.value), float(val.value)

    def frames(self, iq, S, y_t, x_t, alpha, imageOut, do_align=True, want_frames=True, want_raster=False):
        """Context.frames over the group (tsdr_group_frames): frames sharded over the members, combined on the root."""
        z = _c64(iq)
        nb = z.size // int(S)
        if not (isinstance(imageOut, np.ndarray) and imageOut.dtype == np.float32 and imageOut.flags.f_contiguous
                and imageOut.shape == (RENDER_H, RENDER_W)):
            raise AssertionError("imageOut must be a Fortran-order float32 (600,800) array")
        frames = np.empty((nb, RENDER_H, RENDER_W), np.float32) if want_frames else None
        raster = np.empty((nb, int(y_t) * int(x_t)), np.float32) if want_raster else None
        idx = np.zeros((nb, 2), np.int32)
        n = C.c_int(0)
        self._chk(self.lib.tsdr_group_frames(self.h, _ptr(z), z.size, int(S), int(y_t), int(x_t), C.c_float(alpha),
                                             int(bool(do_align)), _ptr(imageOut), _ptr(frames), _ptr(raster), _ptr(idx), C.byref(n)),
                  "tsdr_group_frames")
        out = {"n_frames": n.value, "sync_idx": idx}
        if frames is not None:
            out["frames"] = [frames[f].reshape(-1).reshape((RENDER_H, RENDER_W), order="F") for f in range(nb)]
        if raster is not None:
            out["raster"] = [raster[f].reshape((int(y_t), int(x_t)), order="F") for f in range(nb)]
        return out

    def getWelch(self, fe, sig, sizeFFT=1024, lin=False):
        """Context.getWelch over the group (tsdr_group_welch): segments sharded, one all-reduce of sizeFFT floats."""
        a = np.ascontiguousarray(sig)
        cplx = int(np.iscomplexobj(a))
        a = a.astype(np.complex64 if cplx else np.float32, copy=False)
        y = np.empty(int(sizeFFT), np.float32)
        self._chk(self.lib.tsdr_group_welch(self.h, _ptr(a), cplx, a.size, int(sizeFFT), int(bool(lin)), _ptr(y)), "tsdr_group_welch")
        fAx = (np.arange(int(sizeFFT), dtype=np.float64) / int(sizeFFT) - 0.5) * fe
        return fAx, y


class StagingRing:
    """Pinned-host staging ring: the consumer side of AtomicCircularBuffer / recv!(buffer, csdr)
    (AtomicAbstractSDRs.jl:64-190, 320-322) with the buffer landing on the device.
    fmt "cf32": ComplexF32 slots; "sc16": interleaved int16 I/Q, expanded on the device to ComplexF32 * scale; "sc16raw": int16
    slots that stay int16 on the device (take_d hands out int16 pairs for frames_sc16_d with the same scale)."""

    def __init__(self, ctx, nEch, depth=16, fmt="cf32", scale=1.0):
        self.ctx, self.nEch, self.depth, self.fmt = ctx, int(nEch), int(depth), fmt
        h = C.c_void_p(0)
        ctx.call("tsdr_ring_create", self.nEch, self.depth, {"cf32": 0, "sc16": 1, "sc16raw": 2}[fmt], C.c_float(scale), C.byref(h))
        self.h = h.value

    def _chk(self, rc, what):
        check(self.ctx.h, rc, what)

    def put(self, buf):
        """circ_put!: copy one buffer (complex64[nEch] or int16[2*nEch]) into the ring; never waits for the consumer."""
        a = np.ascontiguousarray(buf)
        want = self.nEch * (8 if self.fmt == "cf32" else 4)
        if a.nbytes != want:
            raise AssertionError(f"ring slot is {want} bytes, got {a.nbytes}")
        self._chk(self.ctx.lib.tsdr_ring_put(self.h, _ptr(a)), "tsdr_ring_put")

    def write_view(self):
        """Zero-copy producer: a numpy view of the pinned slot to fill; publish it with commit()."""
        p = self.ctx.lib.tsdr_ring_write_ptr(self.h)
        n = self.nEch * 2
        ctype = C.c_float if self.fmt == "cf32" else C.c_int16
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(n,))

    def commit(self):
        self._chk(self.ctx.lib.tsdr_ring_commit(self.h), "tsdr_ring_commit")

    def take_d(self, timeout_ms=-1):
        """circ_take! / recv!: device address of the next buffer (nEch ComplexF32); IndexError on timeout/stop."""
        p = C.c_void_p(0)
        self._chk(self.ctx.lib.tsdr_ring_take_d(self.h, int(timeout_ms), C.byref(p)), "tsdr_ring_take_d")
        return p.value

    def stop(self):
        self._chk(self.ctx.lib.tsdr_ring_stop(self.h), "tsdr_ring_stop")

    def stats(self):
        a, b, c = C.c_ulonglong(0), C.c_ulonglong(0), C.c_ulonglong(0)
        rp, rc_ = C.c_double(0), C.c_double(0)
        self._chk(self.ctx.lib.tsdr_ring_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(rp), C.byref(rc_)),
                  "tsdr_ring_stats")
        h, m = C.c_ulonglong(0), C.c_ulonglong(0)
        self._chk(self.ctx.lib.tsdr_ring_prefetch_stats(self.h, C.byref(h), C.byref(m)), "tsdr_ring_prefetch_stats")
        return {"produced": a.value, "consumed": b.value, "overflow": c.value, "producer_msps": rp.value,
                "consumer_msps": rc_.value, "prefetch_hits": h.value, "prefetch_misses": m.value}

    def close(self):
        if self.h:
            self.ctx.lib.tsdr_ring_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SyncXY:
    """SyncXY{Float32} state (FrameSynchronisation.jl:25-48) living on the device."""

    def __init__(self, ctx, y_t, x_t):
        self.ctx = ctx
        self.y_t, self.x_t = int(y_t), int(x_t)
        h = C.c_void_p(0)
        ctx.call("tsdr_sync_create", self.y_t, self.x_t, C.byref(h))
        self.h = h.value
        b = (C.c_int * 4)()
        check(ctx.h, ctx.lib.tsdr_sync_bounds(self.h, b), "tsdr_sync_bounds")
        self.wmin_y, self.wmax_y, self.wmin_x, self.wmax_x = list(b)

    def reset(self):
        check(self.ctx.h, self.ctx.lib.tsdr_sync_reset(self.h), "tsdr_sync_reset")

    def vsync(self, image):
        a = np.asfortranarray(image, dtype=np.float32)
        if a.shape != (self.y_t, self.x_t):
            raise AssertionError("image size does not match the SyncXY state")
        sy, sx = C.c_int(0), C.c_int(0)
        check(self.ctx.h, self.ctx.lib.tsdr_vsync(self.h, _ptr(a), C.byref(sy), C.byref(sx)), "tsdr_vsync")
        return sy.value, sx.value

    def beta(self, which):
        """which='x' -> beta_x (W_x, x_t); 'y' -> beta_y (W_y, y_t); Fortran order like the Julia field."""
        if which == "x":
            shape, w = (1 + self.wmax_x - self.wmin_x, self.x_t), 0
        else:
            shape, w = (1 + self.wmax_y - self.wmin_y, self.y_t), 1
        out = np.empty(shape, np.float32, order="F")
        check(self.ctx.h, self.ctx.lib.tsdr_sync_beta(self.h, w, _ptr(out)), "tsdr_sync_beta")
        return out

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.tsdr_sync_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Resampler:
    """The closure init_resampler returns (Resampler.jl:26-62): call it as r(out, inp)."""

    def __init__(self, ctx, bufferSize, upCoeff):
        self.ctx, self.bufferSize, self.upCoeff = ctx, bufferSize, upCoeff
        h = C.c_void_p(0)
        ctx.call("tsdr_resampler_init", bufferSize, upCoeff, C.byref(h))
        self.h = h.value

    def __call__(self, out, inp):
        if not (isinstance(out, np.ndarray) and isinstance(inp, np.ndarray)):
            raise AssertionError("numpy arrays expected")
        if out.dtype != np.float32 or inp.dtype != np.float32:
            raise AssertionError("Type of input should match type used during init (Float32)")  # Resampler.jl:44
        if inp.size != self.bufferSize:
            raise AssertionError(f"Size of input {inp.size} should match size used during init {self.bufferSize}")  # :47
        if out.size < self.bufferSize * self.upCoeff:
            raise IndexError("out too short")
        x = np.ascontiguousarray(inp)
        check(self.ctx.h, self.ctx.lib.tsdr_resampler_run(self.h, _ptr(x), x.size, _ptr(out)), "tsdr_resampler_run")

    def lpf(self):
        H = np.empty(self.bufferSize * self.upCoeff, np.complex64)
        check(self.ctx.h, self.ctx.lib.tsdr_resampler_lpf(self.h, _ptr(H)), "tsdr_resampler_lpf")
        return H

    def lpf64(self):
        """H as the closure applies it: ComplexF64 (Resampler.jl:93-97)"""
        H = np.empty(self.bufferSize * self.upCoeff, np.complex128)
        check(self.ctx.h, self.ctx.lib.tsdr_resampler_lpf64(self.h, _ptr(H)), "tsdr_resampler_lpf64")
        return H

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.tsdr_resampler_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context(0)
    return _default


# module-level functions with the reference's names, bound to the default context
def amDemod(sig): return default_context().amDemod(sig)
def invert_amDemod(sig): return default_context().invert_amDemod(sig)
def fmDemod(sig): return default_context().fmDemod(sig)
def sig_to_image(sig, y_t, x_t): return default_context().sig_to_image(sig, y_t, x_t)
def downgradeImage(image): return default_context().downgradeImage(image)
def naiveResampler(sigOut, sigId, upCoeff): return default_context().naiveResampler(sigOut, sigId, upCoeff)
def init_resampler(T, bufferSize, upCoeff): return default_context().init_resampler(T, bufferSize, upCoeff)
def calculate_autocorrelation(x, Fs, minDelay, maxDelay, scale="log"):
    return default_context().calculate_autocorrelation(x, Fs, minDelay, maxDelay, scale)
def zoom_autocorr(G, Fs, rate_min=20, rate_max=100): return default_context().zoom_autocorr(G, Fs, rate_min, rate_max)
def getSpectrum(fs, sig, N=None): return default_context().getSpectrum(fs, sig, N)
def getWelch(fe, sig, sizeFFT=1024): return default_context().getWelch(fe, sig, sizeFFT)
def getWaterfall(fe, sig, sizeFFT=1024): return default_context().getWaterfall(fe, sig, sizeFFT)
def vsync(image, sync): return sync.vsync(image)
