#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
BENCH_ARGS="--no-pipeline-leg" bash tools/ab.sh cur ab/colsumdpp.so cur ab/colsumdpp.so > $O/r05_t10_ab.log 2>&1
cat $O/r05_t10_ab.log
for wl in C3 C5; do BENCH_ARGS="--no-pipeline-leg --workload $wl --steps 10" bash tools/ab.sh cur ab/colsumdpp.so 2>&1 | sed "s/^/$wl /"; done | tee -a $O/r05_t10_ab.log
timeout 900 python3 -m pytest tests/test_fast_mode_gpu.py tests/test_full_buffers_fast_gpu.py tests/test_fuzz_gpu.py tests/test_sc16_gpu.py -x -q -m gpu -s 2>&1 | grep -i "worst\|passed\|failed\|error" | tail -24
DUMMY_STREAMS=0 timeout 300 python3 tools/time_pipeline.py 300 C2 2>&1 | grep "pipeline=\|measured"
