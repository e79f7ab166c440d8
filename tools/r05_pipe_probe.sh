#!/bin/bash
# Development aid (GPU box): where does the one-context pipeline lose to two contexts?  Host enqueue time per buffer, the two
# arrangements, hardware-queue count, intra-call split.
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
{
echo "== nproc $(nproc); $(grep -m1 'model name' /proc/cpuinfo)"
echo "== time_pipeline default"; python3 $R/tools/time_pipeline.py 300 C2
echo "== time_pipeline GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 python3 $R/tools/time_pipeline.py 300 C2
echo "== time_pipeline GPU_MAX_HW_QUEUES=8 lanes=3"; GPU_MAX_HW_QUEUES=8 TSDR_PIPE_LANES=3 python3 $R/tools/time_pipeline.py 300 C2
echo "== time_pipeline mode=1 (equal lanes for both)"; TSDR_PIPE_MODE=1 python3 $R/tools/time_pipeline.py 300 C2
echo "== time_pipeline mode=0 (image+tail lanes for both)"; TSDR_PIPE_MODE=0 python3 $R/tools/time_pipeline.py 300 C2
echo "== time_pipeline mode=0 no priority"; TSDR_PIPE_MODE=0 TSDR_PIPE_PRIORITY=0 python3 $R/tools/time_pipeline.py 300 C2
echo "== split call"; python3 $R/tools/time_split_call.py 300 C2
echo "== two contexts"; python3 $R/tools/time_two_contexts.py 2>&1 | tail -8
} > $O/r05_pipe_probe.log 2>&1
tail -50 $O/r05_pipe_probe.log
