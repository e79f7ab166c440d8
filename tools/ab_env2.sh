#!/bin/bash
# Development aid (GPU box): bench.py --quick under several settings of one environment switch, two rounds.
#   tools/ab_env2.sh VAR "v1 v2 v3" [bench args]
VAR=$1; VALS=$2; shift 2
R=${GRAFT_REPO_ROOT:-.}
for i in 1 2; do
for v in $VALS; do
  echo "$VAR=$v: $(env $VAR=$v python3 $R/bench.py --quick --no-pipeline-leg --repeats 5 $@ 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['timing']['ms_per_step_min'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()}, (d.get('index_parity') or {}).get('sync_idx_equal_exact'), (d.get('index_parity') or {}).get('max_rel_pixel_diff_vs_exact'))")"
done; done
