"""BASELINE config 1 as ONE flow (production/investigate_data.jl:30-97,159-206): a synthetic leak written with
writeComplexBinary, read back, and taken through amDemod -> spectrum -> autocorrelation -> refresh rate -> line
count -> mode table -> image -> vsync -> sample offset -> aligned image.  CPU: on the oracle (this is config 1's
"plumbing, no GPU" leg).  GPU: the same flow through the HIP library returns the same picks and offsets."""
import importlib

import numpy as np
import pytest

import oracle_lib as O


class OracleBackend:
    """the oracle behind the product API's method names (what tempestsdr.jl_amd/replay.py calls)"""
    amDemod = staticmethod(O.amDemod)
    calculate_autocorrelation = staticmethod(O.calculate_autocorrelation)
    zoom_autocorr = staticmethod(O.zoom_autocorr)
    sig_to_image = staticmethod(O.sig_to_image)

    @staticmethod
    def getSpectrum(fs, sig, N=None):
        n = len(sig) if N is None else N
        return (np.arange(n) / n - 0.5) * fs, O.getSpectrum(sig, N=N)

    @staticmethod
    def SyncXY(image):
        a = np.asarray(image)
        return O.SyncXY(a.shape[0], a.shape[1])

    @staticmethod
    def vsync(image, sync):
        return sync.vsync(image)


def _capture(tsdr, tmp_path, c1_size=False):
    synth = importlib.import_module("tempestsdr_jl_amd.synth")
    dat = importlib.import_module("tempestsdr_jl_amd.dat_files")
    if c1_size:   # BASELINE config 1 at its own size: a 20 MS/s capture of the 1080p60 mode (2576 x 1125 total), 0.25 s
        Fs, x_t, y_t, fv, n = 20.0e6, 2576, 1125, 60.0, 5_000_000
    else:
        Fs, x_t, y_t, fv, n = 2.0e6, 1056, 628, 60.0, 600_000   # "800x600 @ 60Hz" (VideoConfigurations.jl:27) at 2 MS/s
    iq = synth.synth_leak(Fs, x_t, y_t, fv, n)
    path = str(tmp_path / "dumpIQ_0.dat")
    dat.writeComplexBinary(iq, path, "single")             # DatBinaryFiles.jl:15-31, the GUI's record format
    return path, Fs, (x_t, y_t, fv)


def test_replay_flow_on_oracle(tsdr, tmp_path):
    replay = importlib.import_module("tempestsdr_jl_amd.replay")
    path, Fs, (x_t, y_t, fv) = _capture(tsdr, tmp_path)
    r = replay.replay_file(OracleBackend, path, Fs, offset=42_000)
    assert abs(r["fv"] - fv) < 0.2, r["fv"]
    # the line lag is an integer number of samples (53.08 at this rate): the estimate sits within one lag of the truth
    assert abs(Fs / (r["fv"] * r["y_t"]) - Fs / (fv * y_t)) <= 1.5, r["y_t"]
    assert r["mode"].width == x_t and r["name"] == "800x600 @ 60Hz", (r["name"], r["mode"])
    assert r["sync"][0] == 1                               # stale s_y of a fresh SyncXY (FrameSynchronisation.jl:66)
    assert 1 <= r["sync"][1] <= x_t
    assert r["tau"] == r["sync"][1] * r["mode"].width + r["sync"][0]
    assert r["sample_offset"] == int(np.floor(r["tau"] / (r["mode"].width * r["mode"].height) / r["fv"] * Fs))
    assert r["aligned"].shape == (r["mode"].height, r["mode"].width) and r["aligned"].flags.f_contiguous
    f, y = r["spectrum"]
    assert f.size == y.size == 80_000 and np.isfinite(y).all()
    # the short capture is a BoundsError in the reference, an IndexError here
    with pytest.raises(IndexError):
        dat = importlib.import_module("tempestsdr_jl_amd.dat_files")
        replay.replay(OracleBackend, dat.readComplexBinary(path, "single")[:250_000], Fs, offset=240_000)


@pytest.mark.gpu
def test_replay_flow_gpu_matches_oracle(ctx, tsdr, tmp_path):
    replay = importlib.import_module("tempestsdr_jl_amd.replay")
    path, Fs, _ = _capture(tsdr, tmp_path)
    g = replay.replay_file(ctx, path, Fs, offset=42_000)
    o = replay.replay_file(OracleBackend, path, Fs, offset=42_000)
    for k in ("fv", "lag", "name", "sync", "tau", "sample_offset"):
        assert g[k] == o[k], (k, g[k], o[k])
    assert g["mode"].width == o["mode"].width and g["mode"].height == o["mode"].height
    # the per-function API is EXACT: images bit for bit
    assert np.array_equal(g["image"].view(np.uint32), o["image"].view(np.uint32))
    assert np.array_equal(g["aligned"].view(np.uint32), o["aligned"].view(np.uint32))
    assert np.max(np.abs(g["G"] - o["G"])) < 2e-4


def _check_c1(r, Fs, x_t, y_t, fv):
    assert abs(r["fv"] - fv) < 0.1, r["fv"]
    assert abs(Fs / (r["fv"] * r["y_t"]) - Fs / (fv * y_t)) <= 1.5, r["y_t"]
    assert r["name"] == "1920x1080 @ 60Hz" and r["mode"].width == x_t, (r["name"], r["mode"])
    assert r["sync"][0] == 1 and 1 <= r["sync"][1] <= x_t
    assert r["tau"] == r["sync"][1] * r["mode"].width + r["sync"][0]
    assert r["aligned"].shape == (r["mode"].height, r["mode"].width)


def test_replay_flow_at_c1_size_on_oracle(tsdr, tmp_path):
    """the same flow at BASELINE config 1's own size (production/investigate_data.jl works on a 20 MS/s capture):
    40 MB .dat, n = 4e6 autocorrelation, 1125 x 2576 rasters, vsync on the full raster -- on the oracle"""
    replay = importlib.import_module("tempestsdr_jl_amd.replay")
    path, Fs, (x_t, y_t, fv) = _capture(tsdr, tmp_path, c1_size=True)
    _check_c1(replay.replay_file(OracleBackend, path, Fs, offset=420_000), Fs, x_t, y_t, fv)   # the script's own offset (:159)


@pytest.mark.gpu
def test_replay_flow_at_c1_size_gpu_matches_oracle(ctx, tsdr, tmp_path):
    replay = importlib.import_module("tempestsdr_jl_amd.replay")
    path, Fs, (x_t, y_t, fv) = _capture(tsdr, tmp_path, c1_size=True)
    g = replay.replay_file(ctx, path, Fs, offset=420_000)
    o = replay.replay_file(OracleBackend, path, Fs, offset=420_000)
    _check_c1(g, Fs, x_t, y_t, fv)
    for k in ("fv", "lag", "name", "sync", "tau", "sample_offset"):
        assert g[k] == o[k], (k, g[k], o[k])
    assert g["mode"].width == o["mode"].width and g["mode"].height == o["mode"].height
    assert np.array_equal(g["image"].view(np.uint32), o["image"].view(np.uint32))
    assert np.array_equal(g["aligned"].view(np.uint32), o["aligned"].view(np.uint32))
    assert np.max(np.abs(g["G"] - o["G"])) < 2e-4
