#!/bin/bash
# Development aid (GPU box): cost of the sync guard when no frame is flagged -- bench legs with the guard on / off,
# alternating on one box.   tools/ab_guard.sh [bench args]
cd /tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in 20000 0; do
  echo "TSDR_SYNC_GUARD_PPB=$v: $(env TSDR_SYNC_GUARD_PPB=$v python3 $R/bench.py --quick --repeats 7 $@ | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['timing']['ms_per_step_min'], 'fused', d['fused']['ms_per_step'], d['roofline']['kernels_ms_per_step'])")"
done; done
