"""Golden fixtures (tests/golden/hotpath_v1.npz, RESTATEMENT-GENERATED -- see make_golden.py).

CPU: the oracle still reproduces them bit for bit (regression pin for the checker).
GPU: the HIP path reproduces them (bit-exact on the frame path, stated tolerance on FFT paths)."""
import os
import zlib

import numpy as np
import pytest

import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_v1.npz"))


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def same(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def check_suite(B, sync_cls, frames_fn, exact_fft):
    z = G["iq"]
    assert same(B.amDemod(z), G["am"]) and same(B.invert_amDemod(z), G["inv_am"]) and same(B.abs2(z), G["abs2"])
    assert np.max(np.abs(B.fmDemod(z) - G["fm"])) <= 1e-6
    x = G["rs_in"]
    assert same(B.imresize1d(x, 2898), G["rs_up"]) and same(B.imresize1d(x, 41), G["rs_down"])
    assert same(B.sig_to_image(x, 30, 40), G["s2i"])
    assert same(B.imresize2d(G["img_in"], (20, 30)), G["img_20x30"])
    big = B.downgradeImage(G["img_in"])
    assert crc(big) == G["down_crc"] and same(big.ravel(order="F")[::997], G["down_sub"])
    assert same(B.fill_beta(G["beta_cv"], 101, 3, 25), G["beta"])
    s = sync_cls(77, 131)
    img = G["vs_img"]
    idx = [s.vsync(img), s.vsync(img), s.vsync(np.asfortranarray(np.roll(img, (7, 11), (0, 1))))]
    assert np.array_equal(np.array(idx, np.int32), G["vs_idx"])
    # FFT-based functions
    db, _ = B.calculate_autocorrelation(G["ac_x"], 30000.0, 0.0, 0.05)
    lin, _ = B.calculate_autocorrelation(G["ac_x"], 30000.0, 0.001, 0.05, "lin")
    if exact_fft:
        assert same(db, G["ac_db"]) and same(lin, G["ac_lin"])
    else:
        assert np.max(np.abs(db - G["ac_db"])) < 2e-4 and np.max(np.abs(lin - G["ac_lin"])) < 4e-5 * G["ac_lin"].max()
    for t in ("A", "B"):
        S, y_t, x_t, nfr = [int(v) for v in G[f"fr{t}_geom"]]
        st = np.zeros((600, 800), np.float32, order="F")
        o = frames_fn(G[f"fr{t}_iq"], S, y_t, x_t, st)
        assert o["n_frames"] == nfr and np.array_equal(o["sync_idx"], G[f"fr{t}_idx"])
        assert crc(st) == G[f"fr{t}_state_crc"] and same(st.ravel(order="F")[::499], G[f"fr{t}_state_sub"])
        assert [crc(f) for f in o["frames"]] == list(G[f"fr{t}_frame_crc"])
        assert [crc(f) for f in o["raster"]] == list(G[f"fr{t}_raster_crc"])


def test_oracle_reproduces_golden():
    def frames_fn(iq, S, y_t, x_t, st):
        return O.frames(O.SyncXY(600, 800), iq, S, y_t, x_t, np.float32(0.1), st, want_raster=True)
    check_suite(O, O.SyncXY, frames_fn, exact_fft=True)
    assert same(O.getSpectrum(G["sp_x"], N=1000, lin=True), G["sp_lin"])
    assert same(O.getWelch(G["sp_x"], sizeFFT=256, lin=True), G["welch_lin"])
    assert np.array_equal(O.getWaterfall(G["sp_x"], sizeFFT=128), G["wf"])
    r = O.Resampler(125, 4)
    out = np.empty(500, np.float32)
    r(out, G["up_in"])
    assert same(out, G["up_out"])


@pytest.mark.gpu
def test_hip_reproduces_golden(ctx, tsdr):
    def frames_fn(iq, S, y_t, x_t, st):
        return ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), st, want_raster=True)
    ctx.set_precision("exact")  # the frame-path fixtures are compared bit for bit
    try:
        check_suite(ctx, lambda h, w: tsdr.SyncXY(ctx, h, w), frames_fn, exact_fft=False)
    finally:
        ctx.set_precision("fast")
    _, y = ctx.getSpectrum(1.0, G["sp_x"], N=1000, lin=True)
    assert np.max(np.abs(np.sqrt(y) - np.sqrt(G["sp_lin"]))) < 1e-5 * np.sqrt(G["sp_lin"].max())
    _, y = ctx.getWelch(1.0, G["sp_x"], sizeFFT=256, lin=True)
    assert np.max(np.abs(y - G["welch_lin"])) < 2e-5 * G["welch_lin"].max()
    _, _, m = ctx.getWaterfall(1.0, G["sp_x"], sizeFFT=128)
    assert np.max(np.abs(np.sqrt(m) - np.sqrt(G["wf"]))) < 2e-5 * np.sqrt(G["wf"].max())
    r = ctx.init_resampler(np.float32, 125, 4)
    out = np.empty(500, np.float32)
    r(out, G["up_in"])
    assert np.max(np.abs(out - G["up_out"])) < 1e-5 * np.abs(G["up_out"]).max()
    assert np.max(np.abs(r.lpf() - G["up_H"])) < 2e-5
