"""Dependency-free exchange format between this repo and a Julia machine (tests/golden/make_golden.jl).

One array per file, raw little-endian, named  <name>.<dtype>.<d0>[x<d1>].bin  with dtype in
{f32, f64, c64, i32, i64, u64}.  Matrices are stored COLUMN-MAJOR (Julia's native layout), so the Julia side
is `read!(io, Array{T}(undef, dims...))` and nothing else -- no NPZ/HDF5 package needed.
"""
import os

import numpy as np

_DT = {"f32": np.float32, "f64": np.float64, "c64": np.complex64, "i32": np.int32, "i64": np.int64, "u64": np.uint64}
_NAME = {np.dtype(v): k for k, v in _DT.items()}


def save(directory, name, arr):
    a = np.asarray(arr)
    if a.dtype == np.uint32:
        a = a.astype(np.uint64)
    if a.ndim == 0:
        a = a.reshape(1)
    tag = _NAME[a.dtype]
    dims = "x".join(str(d) for d in a.shape)
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, f"{name}.{tag}.{dims}.bin"), "wb") as f:
        f.write(np.asfortranarray(a).tobytes(order="F"))


def load(directory):
    out = {}
    for fn in sorted(os.listdir(directory)):
        if not fn.endswith(".bin"):
            continue
        name, tag, dims, _ = fn.rsplit(".", 3)
        shape = tuple(int(d) for d in dims.split("x"))
        a = np.fromfile(os.path.join(directory, fn), dtype=_DT[tag])
        out[name] = a.reshape(shape, order="F")
    return out


def chk64(a):
    """Position-sensitive checksum of the 32-bit patterns of an array (column-major order):
    sum_i bits[i] * (2i+1) mod 2^64, i = 0-based.  Julia twin: chk64 in make_golden.jl."""
    b = np.ascontiguousarray(np.asarray(a, np.float32).ravel(order="F")).view(np.uint32).astype(np.uint64)
    w = np.arange(b.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    with np.errstate(over="ignore"):
        return np.uint64(np.sum(b * w, dtype=np.uint64))
