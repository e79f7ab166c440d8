"""Development aid (GPU box): C-ABI misuse.  Every entry point is called with one argument at a time replaced by a bad value
(null pointer, zero / one / absurd size, zero / negative / huge dimension, NaN / negative / huge rate ...).  The bar: the
call RETURNS (any status) -- no crash, no hang -- and the unmodified call still succeeds afterwards.  Sizes are only blown up
on the host-pointer entry points (the library sizes its own staging there and must fail with TSDR_ENOMEM / EINVAL); for the
_d entry points the extent of a device buffer is the caller's contract."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tempest_loader import load_package
T = load_package()
lib = T._lib.load()
ctx = T.Context()
H = ctx.h
rng = np.random.default_rng(3)
keep = []   # keeps host arrays alive


def hbuf(n, dtype=np.float32):
    a = np.ascontiguousarray(rng.random(max(n, 1)).astype(dtype) + 0.1)
    keep.append(a)
    return ("hp", a.ctypes.data)


def dbuf(nbytes):
    p = ctx.dev_alloc(max(nbytes, 16))
    lib.tsdr_upload(H, C.c_void_p(p), (C.c_char * max(nbytes, 16))(), max(nbytes, 16))
    return ("dp", p)


def outv(ctype):
    v = ctype()
    keep.append(v)
    return ("op", C.addressof(v))


sz = lambda v: ("sz", v)
szd = lambda v: ("szd", v)       # a size on a _d entry point: never enlarged
i_ = lambda v: ("i", v)
fl = lambda v: ("fl", v)
f64 = lambda v: ("f64", v)
f32 = lambda v: ("f32", v)
hnd = lambda v: ("hnd", v)

sync_h = C.c_void_p(0); assert lib.tsdr_sync_create(H, 64, 80, C.byref(sync_h)) == 0
sync6 = C.c_void_p(0); assert lib.tsdr_sync_create(H, 600, 800, C.byref(sync6)) == 0
res_h = C.c_void_p(0); assert lib.tsdr_resampler_init(H, 256, 4, C.byref(res_h)) == 0
ring_h = C.c_void_p(0); assert lib.tsdr_ring_create(H, 4096, 4, 0, C.c_float(1.0), C.byref(ring_h)) == 0
N = 4096
S, y_t, x_t, nfr = 980, 70, 130, 2
NPX = 600 * 800
CTX = ("ctx", H)
CASES = {
    "tsdr_am_demod": [CTX, hbuf(2 * N), sz(N), hbuf(N)],
    "tsdr_am_demod_d": [CTX, dbuf(8 * N), szd(N), dbuf(4 * N)],
    "tsdr_invert_am": [CTX, hbuf(2 * N), sz(N), hbuf(N)],
    "tsdr_invert_am_d": [CTX, dbuf(8 * N), szd(N), dbuf(4 * N)],
    "tsdr_fm_demod": [CTX, hbuf(2 * N), sz(N), hbuf(N)],
    "tsdr_fm_demod_d": [CTX, dbuf(8 * N), szd(N), dbuf(4 * N)],
    "tsdr_abs2": [CTX, hbuf(2 * N), sz(N), hbuf(N)],
    "tsdr_resize1d": [CTX, hbuf(N), sz(N), sz(1000), hbuf(1000)],
    "tsdr_resize1d_d": [CTX, dbuf(4 * N), szd(N), szd(1000), dbuf(4000)],
    "tsdr_sig_to_image": [CTX, hbuf(S), sz(S), i_(y_t), i_(x_t), hbuf(y_t * x_t)],
    "tsdr_sig_to_image_d": [CTX, dbuf(4 * S), szd(S), ("id", y_t), ("id", x_t), dbuf(4 * y_t * x_t)],
    "tsdr_resize2d": [CTX, hbuf(64 * 80), i_(64), i_(80), i_(40), i_(50), hbuf(2000)],
    "tsdr_downgrade": [CTX, hbuf(y_t * x_t), i_(y_t), i_(x_t), hbuf(NPX)],
    "tsdr_naive_resample": [CTX, hbuf(N), sz(N), i_(3), hbuf(3 * N)],
    "tsdr_resampler_run": [hnd(res_h.value), hbuf(256), sz(256), hbuf(1024)],
    "tsdr_resampler_run_d": [hnd(res_h.value), dbuf(1024), szd(256), dbuf(4096)],
    "tsdr_resampler_lpf": [hnd(res_h.value), hbuf(2048)],
    "tsdr_resampler_lpf64": [hnd(res_h.value), hbuf(2048, np.float64)],
    "tsdr_autocorr": [CTX, hbuf(N), sz(N), f64(1e4), f64(0.0), f64(0.2), fl(1), hbuf(N), outv(C.c_size_t)],
    "tsdr_autocorr_d": [CTX, dbuf(4 * N), szd(N), f64(1e4), f64(0.0), f64(0.2), fl(1), dbuf(4 * N), outv(C.c_size_t)],
    "tsdr_autocorr_iq_d": [CTX, dbuf(8 * N), szd(N), f64(1e4), f64(0.0), f64(0.2), fl(1), dbuf(4 * N), outv(C.c_size_t)],
    "tsdr_autocorr_search_d": [CTX, dbuf(8 * N), fl(1), szd(N), f64(1e4), f64(0.0), f64(0.2), fl(1), dbuf(4 * N), outv(C.c_size_t),
                               szd(100), szd(500), outv(C.c_size_t), outv(C.c_float)],
    "tsdr_autocorr_partial_d": [CTX, dbuf(4 * N), fl(0), szd(N), szd(0), szd(N // 2), szd(1000), dbuf(4000)],
    "tsdr_autocorr_finish_d": [CTX, dbuf(4 * N), szd(0), szd(1000), fl(1), dbuf(4000)],
    "tsdr_zoom_bounds": [sz(N), f64(1e4), f64(20.0), f64(100.0), outv(C.c_size_t), outv(C.c_size_t)],
    "tsdr_argmax_d": [CTX, dbuf(4 * N), szd(N), outv(C.c_size_t), outv(C.c_float)],
    "tsdr_spectrum": [CTX, hbuf(2 * N), fl(1), sz(N), fl(0), hbuf(N)],
    "tsdr_welch": [CTX, hbuf(2 * N), fl(1), sz(N), sz(256), fl(0), hbuf(256)],
    "tsdr_waterfall": [CTX, hbuf(2 * N), fl(1), sz(N), sz(256), hbuf(N, np.float64)],
    "tsdr_fft_c2c": [CTX, hbuf(2 * N), hbuf(2 * N), sz(N), sz(1), i_(-1)],
    "tsdr_fft_z2z": [CTX, hbuf(2 * 300, np.float64), hbuf(2 * 300, np.float64), sz(300), i_(1)],
    "tsdr_fft_plan": [sz(2_000_000), hbuf(8), i_(8)],
    "tsdr_sync_create": [CTX, i_(64), i_(80), outv(C.c_void_p)],
    "tsdr_sync_reset": [hnd(sync_h.value)],
    "tsdr_sync_bounds": [hnd(sync_h.value), hbuf(4, np.int32)],
    "tsdr_vsync": [hnd(sync_h.value), hbuf(64 * 80), outv(C.c_int), outv(C.c_int)],
    "tsdr_vsync_d": [hnd(sync_h.value), dbuf(4 * 64 * 80), dbuf(8)],
    "tsdr_sync_beta": [hnd(sync_h.value), fl(0), hbuf(80 * 80)],
    "tsdr_fill_beta": [CTX, hbuf(101), ("id", 101), ("id", 3), ("id", 25), hbuf(26 * 101)],   # (room for w_min = 0 / 1)
    "tsdr_circshift_neg": [CTX, hbuf(64 * 80), ("id", 64), ("id", 80), i_(5), i_(7), hbuf(64 * 80)],
    "tsdr_frames": [CTX, hnd(sync6.value), hbuf(2 * S * nfr), sz(S * nfr), ("sz0", S), i_(y_t), i_(x_t), f32(0.1), fl(1), hbuf(NPX),
                    hbuf(nfr * NPX), hbuf(nfr * y_t * x_t), hbuf(2 * nfr, np.int32), outv(C.c_int)],
    "tsdr_frames_d": [CTX, hnd(sync6.value), dbuf(8 * S * nfr), szd(S * nfr), ("sz0", S), ("id", y_t), ("id", x_t), f32(0.1), fl(1),
                      dbuf(4 * NPX), dbuf(4 * nfr * NPX), dbuf(4 * nfr * y_t * x_t), dbuf(8 * nfr), outv(C.c_int)],
    "tsdr_frames_submit_d": [CTX, hnd(sync6.value), dbuf(8 * S * nfr), szd(S * nfr), ("sz0", S), ("id", y_t), ("id", x_t), f32(0.1), fl(1),
                             dbuf(4 * NPX), dbuf(4 * nfr * NPX), dbuf(4 * nfr * y_t * x_t), dbuf(8 * nfr), outv(C.c_int)],
    "tsdr_frames_flush": [CTX],
    "tsdr_frames_scan_d": [CTX, hnd(sync6.value), dbuf(8 * S * nfr), szd(S * nfr), ("sz0", S), ("id", y_t), ("id", x_t), fl(1),
                           dbuf(4 * nfr * NPX), dbuf(4 * nfr * y_t * x_t), dbuf(16 * nfr), outv(C.c_int)],
    "tsdr_ring_create": [CTX, sz(4096), i_(4), fl(0), f32(1.0), outv(C.c_void_p)],
    "tsdr_ring_put": [hnd(ring_h.value), hbuf(2 * 4096)],
    "tsdr_ring_stats": [hnd(ring_h.value), outv(C.c_ulonglong), outv(C.c_ulonglong), outv(C.c_ulonglong), outv(C.c_double), outv(C.c_double)],
    "tsdr_set_option": [CTX, ("str", b"ac_mixed"), fl(1)],
    "tsdr_set_precision": [CTX, fl(1)],
    "tsdr_sync_guard_stats": [CTX, outv(C.c_ulonglong), outv(C.c_ulonglong), fl(0)],
    "tsdr_sync_guard_auto": [CTX, outv(C.c_int), outv(C.c_ulonglong), outv(C.c_ulonglong)],
    "tsdr_sync_guard_margins": [CTX, i_(4), hbuf(8), outv(C.c_int)],
    "tsdr_device_info": [CTX, hbuf(64), sz(256), outv(C.c_int), outv(C.c_size_t)],
}


def conv(spec):
    k, v = spec
    if k in ("ctx", "hp", "dp", "op", "hnd"):
        return C.c_void_p(v)
    if k in ("sz", "szd", "sz0"):
        return C.c_size_t(v)
    if k in ("i", "id", "fl"):
        return C.c_int(v)
    if k == "f64":
        return C.c_double(v)
    if k == "f32":
        return C.c_float(v)
    if k == "str":
        return C.c_char_p(v)
    raise KeyError(k)


def mutations(spec):
    k, v = spec
    if k in ("ctx", "hp", "dp", "op", "hnd"):
        return [(k, 0)]
    if k == "sz":
        return [(k, 0), (k, 1), (k, 1 << 40)]
    if k == "szd":
        return [(k, 0), (k, 1)]
    if k == "sz0":      # a size whose decrease enlarges the output (samples per frame): only the degenerate value
        return [(k, 0)]
    if k == "i":
        return [(k, 0), (k, -1), (k, 1), (k, 2**31 - 1)]   # (INT_MAX x any other dimension of the baselines exceeds the 288 GB of HBM)
    if k == "id":
        return [(k, 0), (k, -1), (k, 1)]
    if k == "fl":
        return [(k, -1), (k, 7)]
    if k == "f64":
        return [(k, 0.0), (k, -1.0), (k, math.nan), (k, 1e300), (k, math.inf)]
    if k == "f32":
        return [(k, math.nan), (k, -1.0)]
    if k == "str":
        return [(k, b"no_such_option"), (k, None)]
    return []


raw = C.CDLL(T._lib.LIB_PATH)
only = [a for a in sys.argv[1:] if not a.startswith("--")]
argpos = [int(a[6:]) for a in sys.argv[1:] if a.startswith("--arg=")]     # restrict the mutated argument positions (bisecting a crash)
total = 0
for name, base in CASES.items():
    if only and name not in only:
        continue
    fn = getattr(raw, name)
    fn.restype = C.c_int
    rc0 = fn(*[conv(a) for a in base])
    lib.tsdr_frames_flush(H); lib.tsdr_synchronize(H)
    ok = (lambda rc: rc >= 0) if name == "tsdr_fft_plan" else (lambda rc: rc == 0)   # (fft_plan returns the pass count)
    assert ok(rc0), (name, "baseline", rc0, lib.tsdr_last_error(H))
    for pos, spec in enumerate(base):
        if argpos and pos not in argpos:
            continue
        for m in mutations(spec):
            args = list(base)
            args[pos] = m
            print(f"{name} arg{pos} {spec[0]} -> {m[1]!r}", end=" ", flush=True)
            rc = fn(*[conv(a) for a in args])
            lib.tsdr_frames_flush(H)
            rs = lib.tsdr_synchronize(H)
            print("rc", rc, "sync", rs, flush=True)
            assert rs == 0, (name, pos, m, "the stream is broken after the call")
            total += 1
    rc1 = fn(*[conv(a) for a in base])
    lib.tsdr_frames_flush(H); lib.tsdr_synchronize(H)
    assert ok(rc1), (name, "baseline after misuse", rc1, lib.tsdr_last_error(H))
print(f"api misuse: {total} mutated calls over {len(CASES)} entry points returned; every baseline still succeeds")
# handles created above (and by the tsdr_sync_create / tsdr_ring_create cases, whose out-handles the baselines overwrite: those leak,
# deliberately -- a caller that never frees them must still be able to exit)
lib.tsdr_ring_stop(ring_h); lib.tsdr_ring_free(ring_h)
lib.tsdr_resampler_free(res_h)
lib.tsdr_sync_free(sync_h); lib.tsdr_sync_free(sync6)
print("handles freed", flush=True)
ctx.close()
print("context closed", flush=True)
