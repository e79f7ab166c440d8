#!/bin/bash
# Development aid (GPU box): alternate bench.py --quick between library builds on ONE box.
#   tools/ab.sh ab/a.so ab/b.so ...      ("cur" = the in-tree build)
R=${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = cur ]; then unset TSDR_HIP_LIB; else export TSDR_HIP_LIB=$R/$lib; fi
    python3 $R/bench.py --quick $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$lib'.ljust(16), 'ms/step', d['ms_per_step'], 'min', d['timing']['ms_per_step_min'], '|', {k: round(v*1e3,1) for k,v in r['kernels_ms_per_step'].items()}, 'fused', d['fused']['ms_per_step'] if d.get('fused') else None)"
  done
done
