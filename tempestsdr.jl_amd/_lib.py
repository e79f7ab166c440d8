"""ctypes binding of libtempest_hip.so (the C ABI declared in include/tempest_hip.h).

This is the Python twin of julia/TempestHIP.jl: same entry points, same status-code ->
exception mapping.  There is no fallback of any kind: if the shared library is missing or
no HIP device can be opened, loading/creating raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# TSDR_HIP_LIB: load another build of the same library (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get("TSDR_HIP_LIB") or os.path.join(HERE, "libtempest_hip.so")

TSDR_OK, TSDR_EINVAL, TSDR_EBOUNDS, TSDR_ENOMEM, TSDR_EHIP, TSDR_ENODEV = 0, -1, -2, -3, -4, -5
RENDER_H, RENDER_W = 600, 800
EXACT, FAST = 0, 1  # tsdr_precision


class TempestHIPError(RuntimeError):
    pass


_lib = None

c_f = C.POINTER(C.c_float)
c_d = C.POINTER(C.c_double)
c_i = C.POINTER(C.c_int)
c_sz = C.c_size_t
c_szp = C.POINTER(C.c_size_t)
vp = C.c_void_p

# name -> (restype, argtypes).  Pointers that may be host OR device are declared c_void_p.
_SIGS = {
    "tsdr_create": (vp, [C.c_int]),
    "tsdr_destroy": (None, [vp]),
    "tsdr_strerror": (C.c_char_p, [C.c_int]),
    "tsdr_last_error": (C.c_char_p, [vp]),
    "tsdr_version": (C.c_char_p, []),
    "tsdr_set_stream": (C.c_int, [vp, vp]),
    "tsdr_synchronize": (C.c_int, [vp]),
    "tsdr_set_precision": (C.c_int, [vp, C.c_int]),
    "tsdr_get_precision": (C.c_int, [vp]),
    "tsdr_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "tsdr_sync_guard_stats": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]),
    "tsdr_sync_guard_auto": (C.c_int, [vp, c_i, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "tsdr_wait_stats": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "tsdr_debug_hold_stream": (C.c_int, [vp, C.c_int]),
    "tsdr_sync_guard_margins": (C.c_int, [vp, C.c_int, vp, c_i]),
    "tsdr_device_info": (C.c_int, [vp, C.c_char_p, c_sz, c_i, c_szp]),
    "tsdr_dev_alloc": (vp, [vp, c_sz]),
    "tsdr_dev_free": (C.c_int, [vp, vp]),
    "tsdr_upload": (C.c_int, [vp, vp, vp, c_sz]),
    "tsdr_download": (C.c_int, [vp, vp, vp, c_sz]),
    "tsdr_timer_start": (C.c_int, [vp]),
    "tsdr_timer_stop": (C.c_int, [vp, c_d]),
    "tsdr_profile_enable": (C.c_int, [vp, C.c_int]),
    "tsdr_profile_reset": (C.c_int, [vp]),
    "tsdr_profile_count": (C.c_int, [vp]),
    "tsdr_profile_get": (C.c_int, [vp, C.c_int, C.c_char_p, c_sz, c_d, C.POINTER(C.c_longlong)]),
    # Demodulation.jl
    "tsdr_am_demod": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_am_demod_d": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_invert_am": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_invert_am_d": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_fm_demod": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_fm_demod_d": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_abs2": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_abs2_d": (C.c_int, [vp, vp, c_sz, vp]),
    # Resampler.jl
    "tsdr_resize1d": (C.c_int, [vp, vp, c_sz, c_sz, vp]),
    "tsdr_resize1d_d": (C.c_int, [vp, vp, c_sz, c_sz, vp]),
    "tsdr_sig_to_image": (C.c_int, [vp, vp, c_sz, C.c_int, C.c_int, vp]),
    "tsdr_sig_to_image_d": (C.c_int, [vp, vp, c_sz, C.c_int, C.c_int, vp]),
    "tsdr_resize2d": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tsdr_resize2d_d": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tsdr_downgrade": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "tsdr_downgrade_d": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "tsdr_naive_resample": (C.c_int, [vp, vp, c_sz, C.c_int, vp]),
    "tsdr_naive_resample_d": (C.c_int, [vp, vp, c_sz, C.c_int, vp]),
    "tsdr_resampler_init": (C.c_int, [vp, c_sz, C.c_int, C.POINTER(vp)]),
    "tsdr_resampler_run": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_resampler_run_d": (C.c_int, [vp, vp, c_sz, vp]),
    "tsdr_resampler_lpf": (C.c_int, [vp, vp]),
    "tsdr_resampler_lpf64": (C.c_int, [vp, vp]),
    "tsdr_fft_z2z": (C.c_int, [vp, vp, vp, c_sz, C.c_int]),
    "tsdr_resampler_free": (None, [vp]),
    # Autocorrelations.jl
    "tsdr_autocorr": (C.c_int, [vp, vp, c_sz, C.c_double, C.c_double, C.c_double, C.c_int, vp, c_szp]),
    "tsdr_autocorr_d": (C.c_int, [vp, vp, c_sz, C.c_double, C.c_double, C.c_double, C.c_int, vp, c_szp]),
    "tsdr_autocorr_iq_d": (C.c_int, [vp, vp, c_sz, C.c_double, C.c_double, C.c_double, C.c_int, vp, c_szp]),
    "tsdr_autocorr_search_d": (C.c_int, [vp, vp, C.c_int, c_sz, C.c_double, C.c_double, C.c_double, C.c_int, vp, c_szp, c_sz, c_sz,
                                         c_szp, c_f]),
    "tsdr_autocorr_partial_d": (C.c_int, [vp, vp, C.c_int, c_sz, c_sz, c_sz, c_sz, vp]),
    "tsdr_autocorr_finish_d": (C.c_int, [vp, vp, c_sz, c_sz, C.c_int, vp]),
    "tsdr_zoom_bounds": (C.c_int, [c_sz, C.c_double, C.c_double, C.c_double, c_szp, c_szp]),
    "tsdr_argmax_d": (C.c_int, [vp, vp, c_sz, c_szp, c_f]),
    # GetSpectrum.jl
    "tsdr_spectrum": (C.c_int, [vp, vp, C.c_int, c_sz, C.c_int, vp]),
    "tsdr_spectrum_d": (C.c_int, [vp, vp, C.c_int, c_sz, C.c_int, vp]),
    "tsdr_welch": (C.c_int, [vp, vp, C.c_int, c_sz, c_sz, C.c_int, vp]),
    "tsdr_welch_d": (C.c_int, [vp, vp, C.c_int, c_sz, c_sz, C.c_int, vp]),
    "tsdr_waterfall": (C.c_int, [vp, vp, C.c_int, c_sz, c_sz, vp]),
    "tsdr_waterfall_d": (C.c_int, [vp, vp, C.c_int, c_sz, c_sz, vp]),
    "tsdr_fft_c2c": (C.c_int, [vp, vp, vp, c_sz, c_sz, C.c_int]),
    "tsdr_fft_c2c_d": (C.c_int, [vp, vp, vp, c_sz, c_sz, C.c_int]),
    "tsdr_fft_plan": (C.c_int, [c_sz, vp, C.c_int]),
    # FrameSynchronisation.jl
    "tsdr_sync_create": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "tsdr_sync_reset": (C.c_int, [vp]),
    "tsdr_sync_free": (None, [vp]),
    "tsdr_sync_bounds": (C.c_int, [vp, c_i]),
    "tsdr_vsync": (C.c_int, [vp, vp, c_i, c_i]),
    "tsdr_vsync_d": (C.c_int, [vp, vp, vp]),
    "tsdr_sync_beta": (C.c_int, [vp, C.c_int, vp]),
    "tsdr_fill_beta": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "tsdr_circshift_neg": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    # frame loop
    "tsdr_frames": (C.c_int, [vp, vp, vp, c_sz, c_sz, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, c_i]),
    "tsdr_frames_d": (C.c_int, [vp, vp, vp, c_sz, c_sz, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, c_i]),
    "tsdr_frames_submit_d": (C.c_int, [vp, vp, vp, c_sz, c_sz, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, c_i]),
    "tsdr_frames_flush": (C.c_int, [vp]),
    "tsdr_frames_sc16_d": (C.c_int, [vp, vp, vp, C.c_float, c_sz, c_sz, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, c_i]),
    "tsdr_frames_submit_sc16_d": (C.c_int, [vp, vp, vp, C.c_float, c_sz, c_sz, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, c_i]),
    "tsdr_frames_pipeline_info": (C.c_int, [vp, c_i, c_i, c_f, C.c_int, C.c_char_p, c_sz]),
    "tsdr_ring_create": (C.c_int, [vp, c_sz, C.c_int, C.c_int, C.c_float, C.POINTER(vp)]),
    "tsdr_ring_free": (None, [vp]),
    "tsdr_ring_put": (C.c_int, [vp, vp]),
    "tsdr_ring_write_ptr": (vp, [vp]),
    "tsdr_ring_commit": (C.c_int, [vp]),
    "tsdr_ring_take_d": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "tsdr_ring_stop": (C.c_int, [vp]),
    "tsdr_ring_stats": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tsdr_ring_prefetch_stats": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "tsdr_frames_scan_d": (C.c_int, [vp, vp, vp, c_sz, c_sz, C.c_int, C.c_int, C.c_int, vp, vp, vp, c_i]),
    "tsdr_frames_combine_d": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]),
    # one process, several GPUs (RCCL inside the library)
    "tsdr_group_create": (C.c_int, [c_i, C.c_int, C.POINTER(vp)]),
    "tsdr_group_destroy": (None, [vp]),
    "tsdr_group_size": (C.c_int, [vp]),
    "tsdr_group_ctx": (vp, [vp, C.c_int]),
    "tsdr_group_last_error": (C.c_char_p, [vp]),
    "tsdr_group_set_precision": (C.c_int, [vp, C.c_int]),
    "tsdr_group_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "tsdr_group_search": (C.c_int, [vp, vp, C.c_int, c_sz, C.c_double, C.c_double, C.c_double, C.c_int, vp, c_szp, c_sz, c_sz,
                                    c_szp, c_f, C.c_int]),
    "tsdr_group_frames": (C.c_int, [vp, vp, c_sz, c_sz, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, c_i]),
    "tsdr_group_sync_reset": (C.c_int, [vp]),
    "tsdr_group_welch": (C.c_int, [vp, vp, C.c_int, c_sz, c_sz, C.c_int, vp]),
    "tsdr_group_timing": (C.c_int, [vp, c_i, c_d]),
}


def exported_names():
    """Every symbol include/tempest_hip.h declares (kept in sync by tests/test_abi.py)."""
    return sorted(_SIGS)


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
    process cannot both own the GPU (whichever initialises second sees no device), and the dynamic
    loader de-duplicates by SONAME only in load order.  So when torch is installed, import it BEFORE
    dlopen-ing libtempest_hip.so: its runtime then serves both.  Without torch (e.g. under the Julia
    shim) the system ROCm runtime is used.  TSDR_NO_TORCH_PRELOAD=1 skips this."""
    if os.environ.get("TSDR_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load():
    """dlopen the library and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise TempestHIPError(
            f"{LIB_PATH} not found: build it with `python tempestsdr.jl_amd/build.py` "
            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, rc, what=""):
    """Map a tsdr_status to the exception class the reference function would throw."""
    if rc == TSDR_OK:
        return
    lib = load()
    detail = lib.tsdr_last_error(ctx_handle).decode() if ctx_handle else ""
    msg = f"{what}: {lib.tsdr_strerror(rc).decode()}" + (f" [{detail}]" if detail else "")
    if rc == TSDR_EINVAL:
        raise AssertionError(msg)  # @assert / MethodError in the reference
    if rc == TSDR_EBOUNDS:
        raise IndexError(msg)  # BoundsError
    if rc == TSDR_ENOMEM:
        raise MemoryError(msg)
    raise TempestHIPError(msg)
