#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
BENCH_ARGS="--no-pipeline-leg" bash tools/ab.sh cur ab/noprescale.so cur ab/noprescale.so > $O/r05_t9_ab.log 2>&1
cat $O/r05_t9_ab.log
timeout 900 python3 -m pytest tests/test_fast_mode_gpu.py tests/test_full_buffers_fast_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -s 2>&1 | grep -i "worst\|passed\|failed\|error" | tail -30
