"""Offline replay of a recorded capture -- BASELINE config 1's flow, production/investigate_data.jl:30-97,159-206:

    readComplexBinary -> amDemod -> getSpectrum(sig[1:80_000]) -> calculate_autocorrelation(sig, Fs, 0, 1/10)
    -> zoom 50..90 Hz -> findmax -> fv -> line count -> find_closest_configuration -> toImage (= sig_to_image)
    -> SyncXY / vsync on the full raster -> offset of the frame start back into the sample stream -> aligned image.

`backend` is anything exposing the processing API under the reference's names (the product's api.Context on a GPU;
the test-suite passes an adapter over its CPU oracle): amDemod, getSpectrum, calculate_autocorrelation, zoom_autocorr,
sig_to_image, SyncXY.  Host-side arithmetic (peak picks, the mode table, the offset formula) follows the script line by
line, quirks included:
  * the autocorrelation runs on the AMPLITUDE here (the GUI uses the power, GUI.jl:70);
  * fv is rounded to two digits (:62);
  * the first vsync call on a fresh SyncXY returns s_y = 1 (beta_y is read before it is filled,
    FrameSynchronisation.jl:66);
  * tau = tup[2] * width + tup[1] (:200) takes the COLUMN shift times the width plus the ROW shift.
The script then overrides the table lookup with constants of its own capture ("1920x1200 @ 60Hz", height 1235,
:96-97); here the mode found is used, with the measured line count as its height (what the GUI does, GUI.jl:571-574).
"""
import numpy as np

from . import search as _search
from . import video_configurations as vc
from .dat_files import readComplexBinary


def replay(backend, sigRx, Fs, offset=420_000, line_method="gui"):
    """-> dict with every intermediate the script plots or prints.  line_method "script": the script's own 500-lag
    estimate (investigate_data.jl:69-82); "gui": the GUI's (the lags after the frame peak, GUI.jl:491-506)."""
    out = {}
    sigId = backend.amDemod(np.ascontiguousarray(sigRx, np.complex64))                    # :37
    out["spectrum"] = backend.getSpectrum(Fs, sigId[:80_000])                              # :44 (freqAx, dB)
    G, _ = backend.calculate_autocorrelation(sigId, Fs, 0, 1 / 10)                         # :52
    rates_large, G_large = backend.zoom_autocorr(G, Fs, rate_min=50, rate_max=90)          # :55
    pos = int(np.argmax(G_large))                                                          # :60 findmax: first maximum
    fv = float(np.round(1.0 / (1.0 / rates_large[pos]), 2))                               # :61-62
    if line_method == "script":
        y_t, m = _search.estimate_line_count(backend, G, Fs, fv)                           # :69-82
    else:
        y_t, m = _search.estimate_line_count_gui(rates_large, G_large, Fs, rates_large[pos])
    name, mode = next(iter(vc.find_closest_configuration(y_t, fv).items()))                # :92
    final = vc.VideoMode(mode.width, int(round(y_t)), fv)                                  # GUI.jl:572-574
    d = int(np.round(Fs / final.refresh))                                                  # toImage :161
    if offset + 2 * d > sigId.size:
        raise IndexError("capture too short for the image at this offset (BoundsError in the reference)")
    image = backend.sig_to_image(sigId[offset: offset + d], final.height, final.width)     # :179 toImage == sig_to_image
    sync = backend.SyncXY(image)                                                           # :196
    tup = backend.vsync(image, sync)                                                       # :197  (s_y, s_x)
    tau = tup[1] * final.width + tup[0]                                                    # :200
    idx = int(np.floor(tau / (final.width * final.height) / fv * Fs))                      # :201
    aligned = backend.sig_to_image(sigId[offset + idx: offset + idx + d], final.height, final.width)   # :206
    out.update(fv=fv, fv_unrounded=float(rates_large[pos]), lag=m, y_t=float(y_t), name=name, mode=final, sync=tuple(int(v) for v in tup),
               tau=int(tau), sample_offset=idx, image=image, aligned=aligned, G=G)
    return out


def replay_file(backend, path, Fs, fmt="single", **kw):
    return replay(backend, readComplexBinary(path, fmt), Fs, **kw)
