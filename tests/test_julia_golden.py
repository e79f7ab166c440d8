"""Pinning the oracle (and the HIP library) to outputs of the reference itself.

tests/golden/v2/inputs   committed inputs (raw vectors, rawvec.py)
tests/golden/v2/oracle   what oracle/tempest_oracle.c returns for them  -- RESTATEMENT-GENERATED
tests/golden/v2/julia    what TempestSDR.jl returns for them            -- written by tests/golden/make_golden.jl on
                         a machine that has Julia; ABSENT until someone runs it (none exists in the build
                         container), in which case the *_vs_julia tests skip and say so.

Comparison policy (per key): bit-exact wherever the reference fixes the arithmetic; the stated tolerance where it
runs an f32 FFT (FFTW) or a transcendental, and -- only against Julia -- for values downstream of the two
reductions Julia evaluates with @simd re-association (sum(image;dims=1) and sum(c_v): see "summation orders" in
the oracle), which no restatement can reproduce bit for bit on an unknown CPU.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import rawvec  # noqa: E402
import make_vectors_v2 as V2  # noqa: E402
import oracle_lib as O  # noqa: E402

V2DIR = os.path.join(HERE, "golden", "v2")
JULIA = os.path.join(V2DIR, "julia")
HAVE_JULIA = os.path.isdir(JULIA) and any(f.endswith(".bin") for f in os.listdir(JULIA))
NEED_JULIA = pytest.mark.skipif(not HAVE_JULIA, reason="tests/golden/v2/julia absent: run `julia tests/golden/make_golden.jl "
                                "<TempestSDR.jl checkout>` on a machine with Julia (parity stays UNPINNED until then)")

# key prefix -> (kind, tolerance).  kinds: "bits" exact bit patterns; "abs" max |a-b| <= tol; "relmax" max |a-b| <= tol*max|b|;
# "sqrtrel" the same on sqrt (power spectra); "eq" integer equality
POLICY = [
    ("fm", ("abs", 1e-6)),                 # atan2 implementations differ (device libm / glibc / Julia's own)
    ("ac_db", ("abs", 2e-4)), ("zoom_G", ("abs", 2e-4)), ("ac_lin", ("relmax", 4e-5)),
    ("zoom_rates", ("relmax", 1e-15)),
    ("sp_db", ("abs", 2e-3)), ("spz_db", ("abs", 2e-3)), ("welch_db", ("abs", 1e-3)), ("wf", ("sqrtrel", 2e-5)),
    ("up_out", ("relmax", 1e-5)), ("up_H", ("relmax", 2e-5)), ("up6_H", ("relmax", 2e-5)), ("up3_H", ("relmax", 2e-5)),
    ("_idx", ("eq", 0)), ("_chk", ("eq", 0)),
]
# against Julia only: downstream of an @simd reduction (column sums, Sigma = sum(c_v))
OPEN_VS_JULIA = [("vs_cv", ("relmax", 1e-6)), ("vs600_cv", ("relmax", 1e-6)), ("vs_bx", ("relmax", 2e-6)), ("vs_by", ("relmax", 2e-6)),
                 ("vs600_bx_sub", ("relmax", 2e-6)), ("vs600_by_sub", ("relmax", 2e-6)), ("vs600_bx_chk", ("skip", 0)),
                 ("vs600_by_chk", ("skip", 0))]


def policy_for(key, vs_julia):
    if vs_julia:
        for pre, pol in OPEN_VS_JULIA:
            if key.startswith(pre):
                return pol
    for pre, pol in POLICY:
        if key.startswith(pre) or key.endswith(pre):
            return pol
    return ("bits", 0)


def compare(cand, ref, vs_julia, only=None):
    bad, exact_open = [], []
    for k in sorted(ref):
        if k not in cand or (only is not None and k not in only):
            continue
        a, b = np.atleast_1d(np.asarray(cand[k])), np.atleast_1d(np.asarray(ref[k]))
        kind, tol = policy_for(k, vs_julia)
        if kind == "skip":
            continue
        if a.shape != b.shape:
            bad.append(f"{k}: shape {a.shape} vs {b.shape}")
            continue
        if kind == "eq":
            ok = np.array_equal(a.astype(np.int64) if a.dtype.kind in "iu" else a, b.astype(np.int64) if b.dtype.kind in "iu" else b)
            err = 0 if ok else 1
        elif kind == "bits":
            a32, b32 = np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32)
            nb = int(np.count_nonzero((a32 != b32) & ~(np.isnan(a) & np.isnan(b))))
            ok, err = nb == 0, nb
        else:
            a64, b64 = a.astype(np.float64), b.astype(np.float64)
            if kind == "sqrtrel":
                a64, b64 = np.sqrt(a64), np.sqrt(b64)
            d = float(np.max(np.abs(a64 - b64))) if a64.size else 0.0
            scale = float(np.max(np.abs(b64)))
            if k.endswith(("_H_re", "_H_im")):  # real and imaginary part of one complex vector: relative to its largest magnitude
                stem = k[:-3]
                scale = float(np.max(np.hypot(np.asarray(ref[stem + "_re"], np.float64), np.asarray(ref[stem + "_im"], np.float64))))
            lim = tol if kind == "abs" else tol * scale
            ok, err = d <= lim, d
            if vs_julia and ok and any(k.startswith(p) for p, _ in OPEN_VS_JULIA) and d == 0.0:
                exact_open.append(k)
        if not ok:
            bad.append(f"{k}: {kind} tol={tol} got {err}")
    return bad, exact_open


def hip_outputs(ctx, tsdr, inp):
    def frames_fn(iq, S, y_t, x_t, st):
        return ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), st, want_raster=True)
    ctx.set_precision("exact")
    try:
        o = V2.outputs(ctx, inp, lambda h, w: tsdr.SyncXY(ctx, h, w), frames_fn)
    finally:
        ctx.set_precision("fast")
    _, o["sp_db"] = ctx.getSpectrum(1.0, inp["sp_x"], N=1000)
    _, o["spz_db"] = ctx.getSpectrum(1.0, inp["sp_z"])
    _, o["welch_db"] = ctx.getWelch(1.0, inp["sp_x"], sizeFFT=256)
    _, _, o["wf"] = ctx.getWaterfall(1.0, inp["sp_x"], sizeFFT=128)
    r = ctx.init_resampler(np.float32, 125, 4)
    out = np.empty(500, np.float32)
    r(out, inp["up_in"])
    H = r.lpf()
    o["up_out"], o["up_H_re"], o["up_H_im"] = out, H.real.astype(np.float64), H.imag.astype(np.float64)
    for tag, nb, up in V2.NEAR_TIE_LPF:
        Hn = ctx.init_resampler(np.float32, nb, up).lpf64()
        o[f"{tag}_H_re"], o[f"{tag}_H_im"] = np.ascontiguousarray(Hn.real), np.ascontiguousarray(Hn.imag)
    nv = np.empty(375, np.float32)
    ctx.naiveResampler(nv, inp["up_in"], 3)
    o["naive"] = nv
    return o


def test_raw_format_roundtrip(tmp_path):
    a = np.asfortranarray(np.arange(12, dtype=np.float32).reshape(3, 4))
    rawvec.save(str(tmp_path), "m", a)
    rawvec.save(str(tmp_path), "z", np.array([1 + 2j], np.complex64))
    got = rawvec.load(str(tmp_path))
    assert np.array_equal(got["m"], a) and got["m"].shape == (3, 4) and got["z"][0] == 1 + 2j
    # column-major on disk: the second float is element (1,0)
    raw = np.fromfile(os.path.join(str(tmp_path), "m.f32.3x4.bin"), np.float32)
    assert raw[1] == a[1, 0]
    assert rawvec.chk64(np.array([1.0, 2.0], np.float32)) == np.uint64(0x3F800000 * 1 + 0x40000000 * 3)


def test_vs600_image_is_integer_built():
    img = V2.vs600_image()
    assert img.shape == (600, 800) and img.dtype == np.float32
    assert img[0, 0] == 0.0 and img[1, 0] == np.float32(37 / 256 * 0.5) and img[200, 5] == 1.0 and img[5, 379] == 1.0


def test_oracle_reproduces_v2_fixture():
    """Regression pin of the checker: the committed oracle outputs are what the oracle returns today."""
    inp = rawvec.load(os.path.join(V2DIR, "inputs"))
    want = rawvec.load(os.path.join(V2DIR, "oracle"))
    got = V2.oracle_outputs(inp)
    assert set(got) == set(want)
    for k in want:
        g = np.asarray(got[k])
        g = g.astype(np.uint64) if g.dtype == np.uint32 else g
        g = g.reshape(1) if g.ndim == 0 else g
        assert g.dtype == want[k].dtype and g.shape == want[k].shape, k
        assert np.asfortranarray(g).tobytes(order="F") == np.asfortranarray(want[k]).tobytes(order="F"), k


@NEED_JULIA
def test_oracle_vs_julia():
    inp = rawvec.load(os.path.join(V2DIR, "inputs"))
    ref = rawvec.load(JULIA)
    bad, exact_open = compare(V2.oracle_outputs(inp), ref, vs_julia=True)
    print("open-order keys that nevertheless matched Julia bit for bit:", exact_open)
    assert not bad, "oracle differs from the Julia reference:\n  " + "\n  ".join(bad)


@pytest.mark.gpu
def test_hip_vs_oracle_v2(ctx, tsdr):
    inp = rawvec.load(os.path.join(V2DIR, "inputs"))
    ref = rawvec.load(os.path.join(V2DIR, "oracle"))
    bad, _ = compare(hip_outputs(ctx, tsdr, inp), ref, vs_julia=False)
    assert not bad, "HIP differs from the oracle fixture:\n  " + "\n  ".join(bad)


@pytest.mark.gpu
@NEED_JULIA
def test_hip_vs_julia(ctx, tsdr):
    inp = rawvec.load(os.path.join(V2DIR, "inputs"))
    ref = rawvec.load(JULIA)
    bad, _ = compare(hip_outputs(ctx, tsdr, inp), ref, vs_julia=True)
    assert not bad, "HIP differs from the Julia reference:\n  " + "\n  ".join(bad)
