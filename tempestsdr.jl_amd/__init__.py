"""tempestsdr.jl_amd -- MI355X-native IQ -> frame reconstruction behind TempestSDR.jl's API.

The directory name carries a dot, so it cannot be imported with a plain `import`; use

    from tempest_loader import load_package      # repo root
    tsdr = load_package()                        # registers it as `tempestsdr_jl_amd`

Contents: csrc/ (hand-written HIP kernels + the C ABI), _lib.py (ctypes binding),
api.py (host mirror with the reference's function names), synth.py (synthetic leak
generator for tests/bench), video_configurations.py and dat_files.py (the callers'
data formats either side of the hot path), parallel.py (one-process-per-GPU sharding),
julia/TempestHIP.jl (the `ccall` shim the Julia runtime loads).
"""
from . import _lib  # noqa: F401
from ._lib import RENDER_H, RENDER_W, TempestHIPError  # noqa: F401
from .api import (Context, Group, Resampler, StagingRing, SyncXY, amDemod, calculate_autocorrelation, default_context,  # noqa: F401
                  downgradeImage, fmDemod, getSpectrum, getWaterfall, getWelch, init_resampler, invert_amDemod,
                  naiveResampler, sig_to_image, vsync, zoom_autocorr)
