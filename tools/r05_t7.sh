#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
timeout 1200 python3 -m pytest tests/test_sc16_gpu.py tests/test_ring_gpu.py tests/test_fast_mode_gpu.py -x -q -m gpu > $O/r05_t7_pytest.log 2>&1
tail -15 $O/r05_t7_pytest.log
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os, importlib
sys.path.insert(0, os.getcwd())
from tempest_loader import load_package
tsdr = load_package()
synth = importlib.import_module("tempestsdr_jl_amd.synth")
ing = importlib.import_module("tempestsdr_jl_amd.ingest")
import torch
ctx = tsdr.Context(0)
w = synth.WORKLOADS["C2"]
S = synth.samples_per_frame(w["Fs"], w["fv"])
iq = synth.synth_leak(w["Fs"], w["x_t"], w["y_t"], w["fv"], S * 30)
for rep in range(2):
    for fmt in ("cf32", "sc16", "sc16raw"):
        print(ing.bench_ingest(ctx, tsdr, iq, S, w["y_t"], w["x_t"], seconds=1.0, fmt=fmt), flush=True)
PY
