"""Configuration search: refresh rate, line count and video mode from a capture.

Mirrors the caller-side flow around the hot path:
  * extract_configuration -- GUI.jl:49-88: sigCorr = abs2.(IQ) over >= 0.1 s, calculate_autocorrelation(
    sigCorr, Fs, 0, 0.1), zoom_autocorr(.; rate_min=50, rate_max=90), findmax -> fv.
  * estimate_line_count -- production/investigate_data.jl:69-82 (the only automatic y_t search in the
    reference): zoom_autocorr(G, Fs; rate_min=fv, rate_max=fv+0.3), first 500 lags, findmax -> m,
    y_t = 1/(fv * m/Fs).
  * find_mode -- investigate_data.jl:92 / GUI.jl:571-574: find_closest_configuration(y_t, fv); the width comes
    from the table, the height is overridden by the measured y_t.
The heavy part (abs2 + the long-lag autocorrelation) runs on the GPU through the C ABI; the peak picks are a few
hundred thousand floats and stay on the host, as in the reference.
"""
import numpy as np

from . import video_configurations as vc


def extract_configuration(ctx, iq, Fs, delay_rate=0.1, rate_min=50, rate_max=90):
    """-> (rates_refresh, G_refresh, fv, G) with the reference's conventions (incl. the zoom off-by-one)."""
    iq = np.ascontiguousarray(iq)
    index_max = int(np.round(delay_rate * Fs))                      # GUI.jl:60
    if iq.size < index_max:
        raise IndexError("capture shorter than the autocorrelation window (BoundsError in the reference)")
    sig_corr = ctx.abs2(iq)                                         # GUI.jl:70 (power, not amplitude)
    G, _ = ctx.calculate_autocorrelation(sig_corr, Fs, 0, delay_rate)  # GUI.jl:73
    rates, Gz = ctx.zoom_autocorr(G, Fs, rate_min=rate_min, rate_max=rate_max)  # GUI.jl:74
    pos = int(np.argmax(Gz))                                        # findmax: first maximum (GUI.jl:79)
    fv = 1.0 / (1.0 / rates[pos])                                   # GUI.jl:80-81
    return rates, Gz, fv, G


def estimate_line_count(ctx, G, Fs, fv, span_hz=0.3, n_lags=500):
    """investigate_data.jl:69-82.  Returns (y_t estimate as float, lag index m)."""
    _, Gs = ctx.zoom_autocorr(G, Fs, rate_min=fv, rate_max=fv + span_hz)
    Gs = Gs[:n_lags]
    m = int(np.argmax(Gs)) + 1                                      # Julia findmax index (1-based)
    tau = m / Fs
    return 1.0 / (fv * tau), m


def estimate_line_count_gui(rates_refresh, G_refresh, Fs, fv, N=1000):
    """The GUI's line-count selection (GUI.jl:491-506 + :238-240) with the mouse click replaced by argmax:
    posFv = argmin(abs.(rates_refresh[1:end-N] .- fv)); G_yt = G_refresh[posFv .+ (1:N)] are the N lags AFTER
    the frame peak, drawn against (0:N-1)/Fs; the strongest one sits one video line later, and
    delay2yt(tau, fv) = round(1/(fv*tau)).  Returns (y_t, 1-based index of the pick)."""
    rates_refresh = np.asarray(rates_refresh)
    pos_fv = int(np.argmin(np.abs(rates_refresh[: rates_refresh.size - N] - fv))) + 1   # 1-based
    G_yt = np.asarray(G_refresh)[pos_fv: pos_fv + N]                                   # posFv .+ (1:N)
    i = int(np.argmax(G_yt)) + 1
    tau = (i - 1) / Fs                                                                  # rates_yt[i]
    return vc.delay2yt(tau, fv), i


def find_mode(y_t, fv):
    """-> (name, VideoMode(width from the table, height = measured y_t, refresh = fv))"""
    sub = vc.find_closest_configuration(y_t, fv)
    name, mode = next(iter(sub.items()))
    return name, vc.VideoMode(mode.width, int(round(y_t)), float(fv))


def search(ctx, iq, Fs, method="gui"):
    """The whole unknown-configuration search (BASELINE config 4) on one capture.
    method "gui": line count from the lags after the frame peak (what the GUI user clicks, automated);
    method "script": production/investigate_data.jl's 500-lag window before the frame peak (kept for fidelity;
    its window does not start on a line multiple, so its estimate is unreliable -- see tests)."""
    rates, Gz, fv, G = extract_configuration(ctx, iq, Fs)
    if method == "gui":
        y_t, m = estimate_line_count_gui(rates, Gz, Fs, fv)
    else:
        y_t, m = estimate_line_count(ctx, G, Fs, fv)
    name, mode = find_mode(y_t, fv)
    return {"fv": fv, "y_t": y_t, "lag": m, "name": name, "mode": mode}
