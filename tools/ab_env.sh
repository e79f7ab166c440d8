#!/bin/bash
# Development aid (GPU box): alternate two settings of one environment switch on the same box.
#   tools/ab_env.sh VAR A B [bench args]   -> ms_per_step and per-kernel times for VAR=A and VAR=B, three rounds
VAR=$1; A=$2; B=$3; shift 3
cd /tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in $A $B; do
  echo "$VAR=$v: $(env $VAR=$v python3 $R/bench.py --quick --repeats 7 $@ | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['timing']['ms_per_step_min'], d['roofline']['kernels_ms_per_step'])")"
done; done
