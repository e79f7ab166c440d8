#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_fast_mode_gpu.py tests/test_frame_path_gpu.py tests/test_group_gpu.py -x -q -m gpu -k "adaptive or pipeline or submit or guard or group" > $O/r05_t3_pytest.log 2>&1
tail -12 $O/r05_t3_pytest.log
{
for rep in 1 2; do for v in 0 1; do
echo "== TSDR_PIPE_EXT_EVENT=$v"; TSDR_PIPE_EXT_EVENT=$v timeout 300 python3 $R/tools/time_pipeline.py 300 C2 2>&1 | grep -v amdgpu.ids
done; done
echo "== two contexts"; python3 $R/tools/time_two_contexts.py 2>&1 | grep contexts=
} > $O/r05_t3_pipe.log 2>&1
cat $O/r05_t3_pipe.log
