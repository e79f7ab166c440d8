"""The C-ABI shared library loads and exports every symbol include/tempest_hip.h declares.
CPU only: no compute entry point is called (there is no GPU here and no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "tempest_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tsdr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_a_sane_surface():
    syms = header_symbols()
    for must in ("tsdr_am_demod", "tsdr_sig_to_image", "tsdr_downgrade", "tsdr_autocorr", "tsdr_zoom_bounds",
                 "tsdr_spectrum", "tsdr_welch", "tsdr_waterfall", "tsdr_sync_create", "tsdr_vsync", "tsdr_frames",
                 "tsdr_resampler_init", "tsdr_naive_resample", "tsdr_invert_am", "tsdr_fm_demod"):
        assert must in syms
    assert len(syms) >= 60


def test_library_exports_every_declared_symbol(tsdr):
    lib = tsdr._lib.load()  # raises if the .so is missing or a prototype cannot be bound
    raw = C.CDLL(tsdr._lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(raw, s), f"{s} declared in include/tempest_hip.h but not exported"
    assert sorted(tsdr._lib.exported_names()) == header_symbols(), "ctypes table and header drifted apart"
    assert b"gfx950" in lib.tsdr_version()
    assert lib.tsdr_strerror(-2).startswith(b"index out of bounds")


def test_no_device_means_loud_failure_not_fallback(tsdr):
    """In this container there is no GPU: creating a context must FAIL (never fall back)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(tsdr.TempestHIPError):
        tsdr.Context(0)


def test_group_without_devices_fails_loudly_and_loads_rccl_on_demand(tsdr):
    """tsdr_group_create without a usable HIP device returns a status and no handle (no CPU fallback for the multi-GPU split
    either); and the library's one collective comes from librccl itself -- loaded with dlopen at the first group of distinct
    devices (group.hip:rccl_api: the entry points are named in the binary), NOT a NEEDED entry: a single-GPU user of the
    library does not need RCCL installed."""
    import shutil
    import subprocess
    import torch
    lib = tsdr._lib.load()
    if not torch.cuda.is_available():
        h = C.c_void_p(0)
        devs = (C.c_int * 1)(0)
        assert lib.tsdr_group_create(devs, 1, C.byref(h)) != 0 and not h.value
        with pytest.raises(tsdr.TempestHIPError):
            tsdr.Group([0])
    assert lib.tsdr_group_size(None) == tsdr._lib.TSDR_EINVAL
    if shutil.which("readelf"):
        out = subprocess.run(["readelf", "-d", tsdr._lib.LIB_PATH], capture_output=True, text=True).stdout
        assert "librccl" not in out, "librccl must be loaded on demand, not linked"
    blob = open(tsdr._lib.LIB_PATH, "rb").read()
    for name in (b"librccl.so", b"ncclCommInitAll", b"ncclAllReduce", b"ncclSend", b"ncclRecv"):
        assert name in blob, f"libtempest_hip.so must name {name.decode()} (tsdr_group_*: ncclAllReduce of the autocorrelation accumulators)"


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tempestsdr.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".jl")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "libtempest_oracle" not in txt, f"{f} references the oracle"
                assert "orc_" not in txt, f"{f} references an oracle symbol"


def test_zoom_bounds_is_pure_host_logic(tsdr):
    lib = tsdr._lib.load()
    a, b = C.c_size_t(0), C.c_size_t(0)
    assert lib.tsdr_zoom_bounds(2_000_000, 20e6, 50.0, 90.0, C.byref(a), C.byref(b)) == 0
    assert (a.value, b.value) == (222_222, 400_000)  # 177 779 points at C2
    assert lib.tsdr_zoom_bounds(1000, 20e6, 50.0, 90.0, C.byref(a), C.byref(b)) == 0
    assert (a.value, b.value) == (1000, 1000)  # min(.,N) clamp, Autocorrelations.jl:46-47


def test_fft_plan_is_pure_host_logic(tsdr):
    """The pass planner of the FFT engines (host arithmetic): factors multiply back to n, none exceeds 256 except the three-step kernels' 500 / 1000 / 2000 (multi-pass splits of at most 2^22 points), at most six
    passes; lengths with a prime factor above 5 take the Bluestein route (0 passes reported)."""
    import random
    lib = tsdr._lib.load()
    f = (C.c_uint * 8)()
    rnd = random.Random(5)
    sizes = [2, 4, 1000, 1024, 4096, 80_000, 2_000_000, 5_000_000, 20_000_000, 4_194_304, 1 << 24, 3 ** 9, 5 ** 8, 2 * 3 * 5]
    for _ in range(300):
        n = (2 ** rnd.randrange(0, 20)) * (3 ** rnd.randrange(0, 8)) * (5 ** rnd.randrange(0, 7))
        if 2 <= n < 2 ** 31:
            sizes.append(n)
    for n in sizes:
        p = lib.tsdr_fft_plan(n, f, 8)
        assert 1 <= p <= 6, (n, p)
        prod = 1
        for i in range(p):
            assert 2 <= f[i] <= 256 or (f[i] in (500, 1000, 2000) and p > 1 and n <= 1 << 22), (n, list(f)[:p])
            prod *= f[i]
        assert prod == n, (n, list(f)[:p])
    # the search transforms of the workloads: C2 two passes of the three-step kernels (2e6 points stay cache-resident), C5
    # and C3 (5e6 / 2e7 points, streamed from HBM) three and four passes of the two-step kernels
    assert lib.tsdr_fft_plan(2_000_000, f, 8) == 2 and sorted(f[:2]) == [1000, 2000]
    assert lib.tsdr_fft_plan(5_000_000, f, 8) == 3 and max(f[:3]) <= 256
    assert lib.tsdr_fft_plan(20_000_000, f, 8) == 4 and max(f[:4]) <= 256
    for n in (7, 999 * 3, 1_000_003, 2 * 7 * 11):
        assert lib.tsdr_fft_plan(n, f, 8) == 0
