#!/bin/bash
# Development aid (GPU box): does the pipeline's overlap depend on the hardware queues its lanes land on?
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
{
for k in 0 1 2 3 4 5 6 7; do
echo "== DUMMY_STREAMS=$k"; DUMMY_STREAMS=$k python3 $R/tools/time_pipeline.py 200 C2 2>&1 | grep pipeline=True
done
} > $O/r05_queue_probe.log 2>&1
cat $O/r05_queue_probe.log
