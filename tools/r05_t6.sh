#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
{
for rep in 1 2; do for d in 0 1 2 3; do
echo "== PIPE_MODE=2 DEBUG=$d"; TSDR_PIPE_MODE=2 TSDR_PIPE_DEBUG=$d timeout 300 python3 $R/tools/time_pipeline.py 300 C2 2>&1 | grep "pipeline="
done; done
} > $O/r05_t6.log 2>&1
cat $O/r05_t6.log
