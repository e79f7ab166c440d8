#!/bin/bash
# Development aid (GPU box): WRITE_SIZE / FETCH_SIZE of the raster launch under the environment given on the command line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf $R/gpurun_out/pmcw_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_$c -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --search-steps 1 > /dev/null 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/pmcw_$c/run_counter_collection.csv')))
v=sorted(float(r['Counter_Value']) for r in rows if 'k_raster_' in r['Kernel_Name']); v=[x for x in v if x > 0.5*v[-1]]
print('$c', len(v), 'median KB', v[len(v)//2], 'max', v[-1], 'min', v[0])
PY
done
