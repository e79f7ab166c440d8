#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
{
for rep in 1 2; do for v in 0 1; do
echo "== TSDR_GUARD_NOWAIT=$v"; TSDR_GUARD_NOWAIT=$v timeout 300 python3 $R/tools/time_pipeline.py 300 C2 2>&1 | grep "pipeline="
done; done
} > $O/r05_t4.log 2>&1
cat $O/r05_t4.log
