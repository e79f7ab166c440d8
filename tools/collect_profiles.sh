#!/bin/bash
# Evidence run for profiles/ (GPU box): bench lines, rocprofv3 kernel stats of the same command, PMC traffic passes.
# Usage (via gpurun): ./tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-r01_x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
python3 $R/bench.py > $O/${TAG}_bench_c2.json 2> $O/${TAG}_bench_c2.err
python3 $R/bench.py --no-cpu --precision exact > $O/${TAG}_bench_c2_exact.json 2>/dev/null
python3 $R/bench.py --no-cpu --workload C3 > $O/${TAG}_bench_c3.json 2>/dev/null
python3 $R/bench.py --no-cpu --workload C5 > $O/${TAG}_bench_c5.json 2>/dev/null
python3 $R/bench.py --no-cpu --pipeline on > $O/${TAG}_bench_c2_pipeline.json 2>/dev/null
rm -rf $O/${TAG}_stats $O/${TAG}_pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o run -- python3 $R/bench.py --no-cpu > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1
done
find $O/${TAG}_stats -name "*kernel_stats.csv" | head -2
ls $O | grep $TAG
