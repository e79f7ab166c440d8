#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
timeout 1200 python3 -m pytest tests/test_frame_path_gpu.py tests/test_fuzz_gpu.py tests/test_sc16_gpu.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do python3 bench.py --quick --precision exact --no-pipeline-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('exact ms/step', d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['kernels_ms_per_step'].items()})"; done
