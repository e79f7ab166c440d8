#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
{
for rep in 1 2 3; do for m in 1 0; do for v in 0 1; do
echo "== PIPE_MODE=$m EXT_EVENT=$v"; TSDR_PIPE_MODE=$m TSDR_PIPE_EXT_EVENT=$v timeout 300 python3 $R/tools/time_pipeline.py 300 C2 2>&1 | grep "pipeline=True"
done; done; done
} > $O/r05_t5.log 2>&1
cat $O/r05_t5.log
