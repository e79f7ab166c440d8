"""Generate tests/golden/hotpath_v1.npz: small inputs + expected outputs for the hot path.

PROVENANCE: RESTATEMENT-GENERATED.  The expected outputs come from oracle/tempest_oracle.c (the
CPU restatement of the reference), NOT from running TempestSDR.jl -- no Julia runtime exists in
the build container and the reference holds no golden vectors for this path.  The fixture pins
the oracle against regressions and gives the GPU tests fixed vectors; it does not pin the oracle
to Julia (see DESIGN.md "Parity status").

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from tempest_loader import load_package  # noqa: E402

load_package()
import importlib  # noqa: E402

synth = importlib.import_module("tempestsdr_jl_amd.synth")


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def main():
    rng = np.random.default_rng(20251017)
    g = {}
    # Demodulation
    z = (5e-3 * (rng.standard_normal(1000) + 1j * rng.standard_normal(1000))).astype(np.complex64)
    g["iq"] = z
    g["am"] = O.amDemod(z)
    g["inv_am"] = O.invert_amDemod(z)
    g["fm"] = O.fmDemod(z)
    g["abs2"] = O.abs2(z)
    # imresize / sig_to_image / downgrade
    x = rng.random(333, dtype=np.float32)
    g["rs_in"] = x
    g["rs_up"] = O.imresize1d(x, 2898)
    g["rs_down"] = O.imresize1d(x, 41)
    g["s2i"] = O.sig_to_image(x, 30, 40)                      # (30,40) column-major
    small = np.asfortranarray(rng.random((45, 64), dtype=np.float32))
    g["img_in"] = small
    g["img_20x30"] = O.imresize2d(small, (20, 30))
    big = O.downgradeImage(small)
    g["down_crc"] = crc(big)
    g["down_sub"] = np.ascontiguousarray(big.ravel(order="F")[::997])
    # FrameSynchronisation
    cv = (rng.random(101) * 50).astype(np.float32)
    g["beta_cv"] = cv
    g["beta"] = O.fill_beta(cv, 101, 3, 25)
    img = np.full((77, 131), 0.25, np.float32)
    img[20:26, :] = 1.0
    img[:, 50:70] = 1.0
    img += 0.01 * rng.random((77, 131), dtype=np.float32)
    img = np.asfortranarray(img)
    s = O.SyncXY(77, 131)
    g["vs_img"] = img
    g["vs_idx"] = np.array([s.vsync(img), s.vsync(img), s.vsync(np.asfortranarray(np.roll(img, (7, 11), (0, 1))))], np.int32)
    # Autocorrelation / spectrum
    p = (rng.random(3000) ** 2).astype(np.float32) * 1e-5
    g["ac_x"] = p
    g["ac_db"], _ = O.calculate_autocorrelation(p, 30000.0, 0.0, 0.05)
    g["ac_lin"], _ = O.calculate_autocorrelation(p, 30000.0, 0.001, 0.05, "lin")
    sig = rng.standard_normal(2000).astype(np.float32) + 2.0
    g["sp_x"] = sig
    g["sp_lin"] = O.getSpectrum(sig, N=1000, lin=True)
    g["welch_lin"] = O.getWelch(sig, sizeFFT=256, lin=True)
    g["wf"] = O.getWaterfall(sig, sizeFFT=128)
    # init_resampler
    r = O.Resampler(125, 4)
    xin = rng.standard_normal(125).astype(np.float32)
    out = np.empty(500, np.float32)
    r(out, xin)
    g["up_in"], g["up_out"], g["up_H"] = xin, out, r.lpf().astype(np.complex64)
    # frame loop: A = 80x50 raster with S == P (imresize copy path); B = 160x125 raster, 4x upsampling
    for tag, (Fs, x_t, y_t, fv, nfr) in {"A": (0.2e6, 80, 50, 50.0, 3), "B": (0.25e6, 160, 125, 50.0, 3)}.items():
        S = synth.samples_per_frame(Fs, fv)
        iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 17, card="plateau")  # the card the committed vectors were made with
        st = np.zeros((600, 800), np.float32, order="F")
        o = O.frames(O.SyncXY(600, 800), iq, S, y_t, x_t, np.float32(0.1), st, want_raster=True)
        g[f"fr{tag}_iq"] = iq
        g[f"fr{tag}_geom"] = np.array([S, y_t, x_t, nfr], np.int64)
        g[f"fr{tag}_idx"] = o["sync_idx"]
        g[f"fr{tag}_state_crc"] = crc(st)
        g[f"fr{tag}_state_sub"] = np.ascontiguousarray(st.ravel(order="F")[::499])
        g[f"fr{tag}_frame_crc"] = np.array([crc(f) for f in o["frames"]], np.uint32)
        g[f"fr{tag}_raster_crc"] = np.array([crc(f) for f in o["raster"]], np.uint32)
    path = os.path.join(HERE, "hotpath_v1.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
