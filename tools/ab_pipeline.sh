#!/bin/bash
# Development aid (GPU box): the frame loop one call per buffer against the two-lane pipeline, raster and raster-free.
R=${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
  for mode in "--pipeline off" "--pipeline on" "--pipeline off --no-raster" "--pipeline on --no-raster"; do
    python3 $R/bench.py --quick $mode $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$mode'.ljust(28), 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'min', d['timing']['ms_per_step_min'], 'parity', (d.get('index_parity') or {}).get('sync_idx_equal_exact'))"
  done
done
