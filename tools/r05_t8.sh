#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
( time timeout 900 python3 -m pytest tests/test_fuzz_gpu.py -x -q -m gpu -s --durations=8 ) > $O/r05_t8_pytest.log 2>&1
tail -40 $O/r05_t8_pytest.log
