#!/bin/bash
# GPU box: every fuzzer at its full case counts with fresh seeds -- a long run (~15 min); log under gpurun_out/<tag>_fuzz_full.log
TAG=${1:-r05_x}; S0=${2:-777001}
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out; L=$O/${TAG}_fuzz_full.log
: > $L
run() { echo "== $*" >> $L; ( timeout 900 python3 "$@" 2>&1 | grep -v amdgpu.ids | tail -4 ) >> $L; echo "rc=${PIPESTATUS[0]}" >> $L; }
for s in 0 1 2; do run $R/tools/fuzz_frames.py $((S0 + s)) 120; done
for s in 0 1; do run $R/tools/fuzz_raster.py $((S0 + 10 + s)) 100; done
for s in 0 1; do run $R/tools/fuzz_fft.py $((S0 + 20 + s)) 100; done
for s in 0 1; do run $R/tools/fuzz_misc.py $((S0 + 30 + s)); done
run $R/tools/fuzz_api_errors.py
cat $L | tail -60
