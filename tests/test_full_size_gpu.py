"""GPU parity at BASELINE.json's full sizes for the configurations round 1 left unchecked: the C3 and C5
configuration searches (n = 4e7 and 1e7 autocorrelation windows) and one full C3 capture buffer (1e8 IQ samples)
through the frame loop in EXACT mode.  (C2 at full size: test_frame_path_gpu.test_full_c2_buffer_bitexact and
test_fft_path_gpu.)  The oracle needs about a minute per search size on one host core."""
import zlib

import numpy as np
import pytest

import oracle_lib as O
from sync_margin import beta_margin, peak_margin_db

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["C5", "C3"])
def test_search_full_size(ctx, synth, name):
    """extract_configuration (GUI.jl:56-81) at the workload's own size: abs2 -> circular autocorrelation over
    n = 2*round(0.1*Fs) samples -> zoom 50..90 Hz -> findmax.  dB within 2e-4 of the oracle's f64 evaluation over
    every lag, identical argmax in the zoom window; the peak's margin over everything else is printed and must be
    far above the tolerance, so "identical" is not luck."""
    wl = synth.WORKLOADS[name]
    Fs, x_t, y_t, fv = wl["Fs"], wl["x_t"], wl["y_t"], wl["fv"]
    n = 2 * int(round(0.1 * Fs))
    iq = synth.synth_leak(Fs, x_t, y_t, fv, n)
    x = ctx.abs2(iq)
    assert np.array_equal(x.view(np.uint32), O.abs2(iq).view(np.uint32))
    g, _ = ctx.calculate_autocorrelation(x, Fs, 0, 0.1)
    o, _ = O.calculate_autocorrelation(x, Fs, 0, 0.1)
    assert g.size == o.size == n // 2
    err = float(np.max(np.abs(g.astype(np.float64) - o)))
    rg, zg = ctx.zoom_autocorr(g, Fs, rate_min=50, rate_max=90)
    ro, zo = O.zoom_autocorr(o, Fs, rate_min=50, rate_max=90)
    ig, mg = peak_margin_db(zg)
    io, mo = peak_margin_db(zo)
    print(f"{name}: n={n} max|dB err|={err:.2e} zoom argmax gpu={ig} oracle={io} fv={rg[ig]:.4f} Hz "
          f"peak margin gpu={mg:.3f} dB oracle={mo:.3f} dB")
    assert err < 2e-4, err
    assert ig == io and rg[ig] == ro[io]
    assert abs(rg[ig] - fv) < 0.05
    assert min(mg, mo) > 10 * 2e-4, (mg, mo)   # the decision margin is far above the 2e-4 dB the two may differ by


@pytest.mark.parametrize("name", ["C3", "C5"])
def test_full_buffer_bitexact(ctx, tsdr, synth, name):
    """BASELINE configs C3 and C5 at full size -- C3: one 0.5 s buffer at 200 MS/s (1e8 IQ samples, 800 MB, 30 frames of
    2576x1125@60 down-sampled 1.15:1); C5: 0.5 s of 4400x2250@60 (4K60) at 50 MS/s, 30 rasters of 9.9e6 pixels -- through
    tsdr_frames in EXACT mode against the oracle: sync indices, a checksum of every frame, first and last raster and the
    final IIR state bit for bit; the beta margin of the last frame is printed.  (C2: tests/test_frame_path_gpu.py.)"""
    wl = synth.WORKLOADS[name]
    Fs, x_t, y_t, fv = wl["Fs"], wl["x_t"], wl["y_t"], wl["fv"]
    n = int(round(wl["acquisition"] * Fs))
    S = synth.samples_per_frame(Fs, fv)
    iq = synth.synth_leak(Fs, x_t, y_t, fv, n)
    gs = np.zeros((600, 800), np.float32, order="F")
    os_ = np.zeros((600, 800), np.float32, order="F")
    ctx.set_precision("exact")
    try:
        g_sync = tsdr.SyncXY(ctx, 600, 800)
        g = ctx.frames(g_sync, iq, S, y_t, x_t, np.float32(0.1), gs, want_raster=True)
    finally:
        ctx.set_precision("fast")
    o_sync = O.SyncXY(600, 800)
    o = O.frames(o_sync, iq, S, y_t, x_t, np.float32(0.1), os_, want_raster=True)
    assert g["n_frames"] == o["n_frames"] == 30
    assert np.array_equal(g["sync_idx"], o["sync_idx"]), (g["sync_idx"].tolist(), o["sync_idx"].tolist())
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())
    assert [crc(f) for f in g["frames"]] == [crc(f) for f in o["frames"]]
    for f in (0, 29):
        assert np.array_equal(g["raster"][f].view(np.uint32), o["raster"][f].view(np.uint32)), f
    assert np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
    for w in ("x", "y"):
        assert np.array_equal(g_sync.beta(w).view(np.uint32), o_sync.beta(w).view(np.uint32)), w
        print(f"{name} last frame beta_{w}: argmax column, margin = {beta_margin(o_sync.beta(w))}")
