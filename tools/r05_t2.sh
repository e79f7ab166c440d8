#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/r05_t2_pytest.log 2>&1
tail -8 $O/r05_t2_pytest.log
{
for k in 0 5; do
echo "== DUMMY_STREAMS=$k"; DUMMY_STREAMS=$k timeout 300 python3 $R/tools/time_pipeline.py 200 C2 2>&1 | grep -v amdgpu.ids
done
} > $O/r05_t2_pipe.log 2>&1
cat $O/r05_t2_pipe.log
( time python3 bench.py --steps 20 --warmup 5 > $O/r05_t2_bench.json 2> $O/r05_t2_bench.err ) 2>&1 | tail -3
python3 tools/show_bench.py $O/r05_t2_bench.json 2>/dev/null | head -60
