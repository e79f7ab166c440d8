"""DatBinaryFiles.jl mirrored on the host: raw interleaved I,Q `.dat` files,
GNU Radio file-sink compatible (DatBinaryFiles.jl:5).  Formats: "short" (Int16), "single"
(Float32), "double" (Float64) -- :15-31 (write) and :44-66 (read).
"""
import numpy as np

_DT = {"short": np.int16, "single": np.float32, "double": np.float64}


def writeComplexBinary(x, fileID, format="single"):
    x = np.asarray(x)
    if not np.iscomplexobj(x):
        raise AssertionError("expected a complex array (MethodError in the reference)")
    out = np.empty(2 * x.size, _DT["short" if format == "short" else ("single" if format == "single" else "double")])
    re, im = x.real.ravel(), x.imag.ravel()
    if format == "short":
        scale = 1 << 14  # :17-20: each component normalised by ITS OWN maximum, then rounded
        out[0::2] = np.round(scale * re / re.max())
        out[1::2] = np.round(scale * im / im.max())
    else:
        out[0::2] = re
        out[1::2] = im
    out.tofile(fileID)


def readComplexBinary(file, format="single", nbSeg=None):
    if format not in _DT:
        raise ValueError(f"Unsupported format for readComplexBinary. Only support short, single, double and got {format}")
    y = np.fromfile(file, dtype=_DT[format], count=-1 if nbSeg is None else int(nbSeg))
    n = y.size // 2 * 2
    # "always output a ComplexF32" is the reference's intent (:64 comment); Int16/Float64 inputs keep
    # their values, the complex container is what the hot path takes (ComplexF32)
    return (y[0:n:2] + 1j * y[1:n:2]).astype(np.complex64 if format != "double" else np.complex128)
