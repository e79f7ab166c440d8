"""Write tests/golden/v2/inputs (inputs for BOTH sides) and tests/golden/v2/oracle (what the CPU restatement
returns for them), in the raw format of rawvec.py.

    python tests/golden/make_vectors_v2.py            # regenerate inputs + oracle outputs
    julia  tests/golden/make_golden.jl <TempestSDR.jl checkout>    # on a machine with Julia: writes v2/julia

PROVENANCE of v2/oracle: RESTATEMENT-GENERATED (oracle/tempest_oracle.c), not Julia output.  v2/julia, once
someone has run make_golden.jl, IS reference output; tests/test_julia_golden.py then checks the oracle and the
HIP library against it and the parity status in DESIGN.md can move from "unpinned" to "pinned".
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import rawvec  # noqa: E402

V2 = os.path.join(HERE, "v2")
FRAME_CASES = {"A": (0.2e6, 80, 50, 50.0, 3), "B": (0.25e6, 160, 125, 50.0, 3)}  # Fs, x_t, y_t, fv, frames


def vs600_image():
    """600x800 image both languages can build bit for bit from integers (see make_golden.jl vs600_image)."""
    i = np.arange(600, dtype=np.int64)[:, None]
    j = np.arange(800, dtype=np.int64)[None, :]
    img = (((i * 37 + j * 101 + (i * j) % 53) % 256).astype(np.float32) / np.float32(256.0)) * np.float32(0.5)
    img[200:230, :] = 1.0
    img[:, 300:380] = 1.0
    return np.asfortranarray(img)


def make_inputs():
    from tempest_loader import load_package
    load_package()
    import importlib
    synth = importlib.import_module("tempestsdr_jl_amd.synth")
    rng = np.random.default_rng(20251017)
    g = {}
    g["iq"] = (5e-3 * (rng.standard_normal(1000) + 1j * rng.standard_normal(1000))).astype(np.complex64)
    g["rs_in"] = rng.random(333, dtype=np.float32)
    g["img_in"] = np.asfortranarray(rng.random((45, 64), dtype=np.float32))
    g["beta_cv"] = (rng.random(101) * 50).astype(np.float32)
    img = np.full((77, 131), 0.25, np.float32)
    img[20:26, :] = 1.0
    img[:, 50:70] = 1.0
    img += 0.01 * rng.random((77, 131), dtype=np.float32)
    g["vs_img0"] = np.asfortranarray(img)
    g["vs_img1"] = np.asfortranarray(img + 0.01 * rng.random((77, 131), dtype=np.float32))
    g["vs_img2"] = np.asfortranarray(np.roll(img, (7, 11), (0, 1)))
    g["ac_x"] = (rng.random(3000) ** 2).astype(np.float32) * np.float32(1e-5)
    g["sp_x"] = rng.standard_normal(2000).astype(np.float32) + np.float32(2.0)
    g["sp_z"] = (rng.standard_normal(2048) + 1j * rng.standard_normal(2048)).astype(np.complex64)
    g["up_in"] = rng.standard_normal(125).astype(np.float32)
    for tag, (Fs, x_t, y_t, fv, nfr) in FRAME_CASES.items():
        S = synth.samples_per_frame(Fs, fv)
        g[f"fr{tag}_iq"] = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 17, card="plateau")  # the card the committed vectors were made with
        g[f"fr{tag}_geom"] = np.array([S, y_t, x_t, nfr], np.int64)
    return g


def sub(a, step):
    return np.ascontiguousarray(np.asarray(a).ravel(order="F")[::step])


# initLPF sizes (bufferSize, upCoeff) whose round.(exp(im*theta)) has an entry 1e-13 from a tie (sizeFFT = 2436, 2691)
NEAR_TIE_LPF = (("up6", 1218, 2), ("up3", 2691, 1))


def outputs(B, inp, sync_cls, frames_fn, vsync_debug=None):
    """Everything make_golden.jl writes, from backend B (oracle_lib, or the HIP api Context)."""
    o = {}
    z = inp["iq"]
    o["am"], o["inv_am"], o["fm"], o["abs2"] = B.amDemod(z), B.invert_amDemod(z), B.fmDemod(z), B.abs2(z)
    x = inp["rs_in"]
    o["rs_up"], o["rs_down"] = B.imresize1d(x, 2898), B.imresize1d(x, 41)
    o["s2i"] = B.sig_to_image(x, 30, 40)
    o["img_20x30"] = B.imresize2d(inp["img_in"], (20, 30))
    big = B.downgradeImage(inp["img_in"])
    o["down_sub"], o["down_chk"] = sub(big, 997), rawvec.chk64(big)
    o["beta"] = B.fill_beta(inp["beta_cv"], 101, 3, 25)
    s = sync_cls(77, 131)
    idx = []
    for k in range(3):
        img = inp[f"vs_img{k}"]
        if vsync_debug is not None:
            o[f"vs_cv{k}"], o[f"vs_ch{k}"] = vsync_debug(s, img)
        idx.append(s.vsync(img))
        o[f"vs_bx{k}"], o[f"vs_by{k}"] = s.beta("x"), s.beta("y")
    o["vs_idx"] = np.array(idx, np.int32)
    s6 = sync_cls(600, 800)
    img6 = vs600_image()
    if vsync_debug is not None:
        o["vs600_cv"], o["vs600_ch"] = vsync_debug(s6, img6)
    i0 = s6.vsync(img6)
    i1 = s6.vsync(img6)
    o["vs600_idx"] = np.array([i0, i1], np.int32)
    o["vs600_bx_sub"], o["vs600_by_sub"] = sub(s6.beta("x"), 101), sub(s6.beta("y"), 101)
    o["vs600_bx_chk"], o["vs600_by_chk"] = rawvec.chk64(s6.beta("x")), rawvec.chk64(s6.beta("y"))
    o["ac_db"], _ = B.calculate_autocorrelation(inp["ac_x"], 30000.0, 0.0, 0.05)
    o["ac_lin"], _ = B.calculate_autocorrelation(inp["ac_x"], 30000.0, 0.001, 0.05, "lin")
    rates, Gz = B.zoom_autocorr(o["ac_db"], 30000.0, rate_min=25, rate_max=90)
    o["zoom_rates"], o["zoom_G"] = np.asarray(rates, np.float64), np.asarray(Gz, np.float32)
    for tag, (Fs, x_t, y_t, fv, nfr) in FRAME_CASES.items():
        S, y_t, x_t, nfr = [int(v) for v in inp[f"fr{tag}_geom"]]
        st = np.zeros((600, 800), np.float32, order="F")
        r = frames_fn(inp[f"fr{tag}_iq"], S, y_t, x_t, st)
        o[f"fr{tag}_idx"] = np.asarray(r["sync_idx"], np.int32)
        o[f"fr{tag}_frame_chk"] = np.array([rawvec.chk64(f) for f in r["frames"]], np.uint64)
        o[f"fr{tag}_raster_chk"] = np.array([rawvec.chk64(f) for f in r["raster"]], np.uint64)
        o[f"fr{tag}_state_sub"], o[f"fr{tag}_state_chk"] = sub(st, 499), rawvec.chk64(st)
    return o


def oracle_outputs(inp):
    import oracle_lib as O

    def frames_fn(iq, S, y_t, x_t, st):
        return O.frames(O.SyncXY(600, 800), iq, S, y_t, x_t, np.float32(0.1), st, want_raster=True)

    o = outputs(O, inp, O.SyncXY, frames_fn, vsync_debug=lambda s, img: s.project(img))
    # FFT-based rows the HIP api returns with an extra axis: written here in the reference's own form (dB)
    o["sp_db"] = O.getSpectrum(inp["sp_x"], N=1000)
    o["spz_db"] = O.getSpectrum(inp["sp_z"])
    o["welch_db"] = O.getWelch(inp["sp_x"], sizeFFT=256)
    o["wf"] = O.getWaterfall(inp["sp_x"], sizeFFT=128)
    r = O.Resampler(125, 4)
    out = np.empty(500, np.float32)
    r(out, inp["up_in"])
    H = r.lpf()
    o["up_out"], o["up_H_re"], o["up_H_im"] = out, np.ascontiguousarray(H.real), np.ascontiguousarray(H.imag)
    for tag, nb, up in NEAR_TIE_LPF:
        Hn = O.Resampler(nb, up).lpf()
        o[f"{tag}_H_re"], o[f"{tag}_H_im"] = np.ascontiguousarray(Hn.real), np.ascontiguousarray(Hn.imag)
    o["naive"] = O.naiveResampler(inp["up_in"], 3)
    return o


def main():
    inp = make_inputs()
    for d in ("inputs", "oracle"):
        p = os.path.join(V2, d)
        if os.path.isdir(p):
            for fn in os.listdir(p):
                os.remove(os.path.join(p, fn))
    for k, v in inp.items():
        rawvec.save(os.path.join(V2, "inputs"), k, v)
    out = oracle_outputs(rawvec.load(os.path.join(V2, "inputs")))
    for k, v in out.items():
        rawvec.save(os.path.join(V2, "oracle"), k, v)
    size = sum(os.path.getsize(os.path.join(V2, d, f)) for d in ("inputs", "oracle") for f in os.listdir(os.path.join(V2, d)))
    print(f"{len(inp)} inputs, {len(out)} oracle outputs, {size} bytes under {V2}")


if __name__ == "__main__":
    main()
