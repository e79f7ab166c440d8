#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_group_gpu.py tests/test_fast_mode_gpu.py tests/test_frame_path_gpu.py -x -q -m gpu -k "group or pipeline or submit" > $O/r05_t1_pytest.log 2>&1
tail -15 $O/r05_t1_pytest.log
{
for k in 0 1 5 6; do
echo "== DUMMY_STREAMS=$k"; DUMMY_STREAMS=$k timeout 300 python3 $R/tools/time_pipeline.py 200 C2 2>&1 | grep -v amdgpu.ids
done
} > $O/r05_t1_pipe.log 2>&1
cat $O/r05_t1_pipe.log
