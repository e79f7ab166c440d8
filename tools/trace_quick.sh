#!/bin/bash
# Development aid (GPU box): kernel timeline of a short bench run.   tools/trace_quick.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/${TAG}_trace
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace -o run -- python3 $R/bench.py --quick --repeats 1 --steps 6 --warmup 2 --search-steps 1 $@ > /dev/null 2>&1
ls $O/${TAG}_trace
