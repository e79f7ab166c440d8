"""Build libtempest_hip.so (gfx950), in-tree.

    python tempestsdr.jl_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is load-bearing:
the frame path promises the oracle's exact rounding sequence (no fused multiply-adds).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libtempest_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]
# Measurement-only alternatives of two reduction orders (see oracle/tempest_oracle.c "summation orders"; the oracle is
# built with -DORC_ROWSUM_CHUNK8 / -DORC_FIR_NOFMA to match):
#   TSDR_BUILD_DEFINES="TSDR_ROWSUM_CHUNK8 TSDR_FIR_NOFMA" python tempestsdr.jl_amd/build.py --force
FLAGS += ["-D" + d for d in os.environ.get("TSDR_BUILD_DEFINES", "").split()]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "tempest_hip.h"))
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(cc, jobs):
            if verbose:
                print(f"[hipcc] {os.path.basename(src)} rc={rc}")
            if rc != 0:
                sys.stderr.write(out)
                raise RuntimeError(f"hipcc failed on {src}")
            if out.strip() and verbose:
                sys.stderr.write(out)
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        # librccl -- the one collective of the path (tsdr_group_*: all-reduce of the autocorrelation accumulators over xGMI) -- is
        # loaded with dlopen at the first group of distinct devices (group.hip:rccl_api), not linked
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[link] {LIB}")
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
