"""BASELINE config 4: unknown-configuration search on synthetic leaks of three modes from the reference's
table: GPU autocorrelation + the reference's host-side peak picks recover the mode key.
The same flow runs on the oracle to show the two pick the same peaks."""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


class OracleCtx:
    """oracle functions behind the Context method names used by search.py"""
    abs2 = staticmethod(O.abs2)
    calculate_autocorrelation = staticmethod(O.calculate_autocorrelation)
    zoom_autocorr = staticmethod(O.zoom_autocorr)


@pytest.mark.parametrize("name", ["1024x768 @ 60Hz", "1280x1024 @ 75Hz", "1920x1080 @ 60Hz"])
def test_unknown_mode_search(ctx, tsdr, synth, name):
    search = importlib.import_module("tempestsdr_jl_amd.search")
    vc = importlib.import_module("tempestsdr_jl_amd.video_configurations")
    mode = vc.allVideoConfigurations[name]
    Fs = 20e6
    # 0.1 s autocorrelation window needs 2*0.1*Fs samples for the full circular estimate (Autocorrelations.jl:27)
    iq = synth.synth_leak(Fs, mode.width, mode.height, mode.refresh, int(0.2 * Fs))
    got = search.search(ctx, iq, Fs)
    ref = search.search(OracleCtx, iq, Fs)
    # identical picks on GPU and oracle: same lag, same refresh, same table entry
    assert got["lag"] == ref["lag"] and abs(got["fv"] - ref["fv"]) < 1e-9 and got["name"] == ref["name"], (got, ref)
    # the refresh estimate may sit one video line off the frame lag (sub-sample alignment of the bar edges)
    assert abs(got["fv"] - mode.refresh) < mode.refresh / mode.height * 1.5, got
    # line count from the first strong lag after the frame peak (GUI.jl:491-506 automated): integer-lag
    # quantisation and the plot's (i-1)/Fs labelling leave it within ~1 %
    assert abs(got["y_t"] - mode.height) <= 0.02 * mode.height, got
    # the offline script's estimate (investigate_data.jl:69-82) agrees between GPU and oracle too, whatever it is
    gs, rs = search.search(ctx, iq, Fs, method="script"), search.search(OracleCtx, iq, Fs, method="script")
    assert gs["lag"] == rs["lag"] and gs["name"] == rs["name"]
    found = vc.allVideoConfigurations[got["name"]]
    assert found.refresh == mode.refresh and abs(found.height - mode.height) <= 0.02 * mode.height, got
    # where no other mode of that rate lies within 2 % of the height, the key itself must be recovered
    rivals = [k for k, m in vc.allVideoConfigurations.items()
              if m.refresh == mode.refresh and k != name and abs(m.height - mode.height) <= 0.02 * mode.height]
    if not rivals:
        assert got["name"] == name and got["mode"].width == mode.width, got
