#!/bin/bash
# GPU box: the whole GPU suite + smoke, log kept under gpurun_out/<tag>_gputest.log
TAG=${1:-r05_x}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) > $O/${TAG}_gputest.log 2>&1
tail -6 $O/${TAG}_gputest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/${TAG}_gputest.log
