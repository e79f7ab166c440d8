#!/bin/bash
# Round 6 evidence set (GPU box), ONE per final commit, collected with the DRIVER'S arguments:
#   tools/collect_profiles_r06.sh <tag>      -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
#  1. `python3 bench.py --gpus 1 --steps 20 --warmup 5` from the repo root, its wall time beside it;
#  2. rocprofv3 --kernel-trace --stats of the same command's headline legs (`--legs none --no-cpu`: the other legs are child
#     processes of their own; the main leg -- what `roofline` describes -- is the same code, steps and buffers);
#  3. separate --pmc FETCH_SIZE / WRITE_SIZE passes (headline kernels; the raster-free and exact legs' kernels; the search);
#  4. profiles/traffic.json rebuilt (PMC bytes + the profiler's average durations), then the driver's command once more, so
#     that the kept line's `traffic` and `roofline.profiled` are this set's.
TAG=${1:-r06_x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
run_bench() {
  local s=$(date +%s.%N)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
  local rc=$?
  local e=$(date +%s.%N)
  python3 -c "print('rc=$rc wall_s=%.1f' % ($e - $s))" > $O/${TAG}_bench_wall.txt
}
run_bench
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_stats $O/${TAG}_pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o run -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --legs none --no-cpu > $O/${TAG}_bench_under_rocprof.json 2> /dev/null
cp $(find $O/${TAG}_stats -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_bench.csv
rm -rf $O/${TAG}_stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c/main -o run -- python3 $R/bench.py --quick --steps 5 --warmup 2 --repeats 1 > /dev/null 2>&1
  for leg in fused exact search; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c/$leg -o run -- python3 $R/bench.py --child $leg --steps 5 --warmup 2 --repeats 3 --search-steps 3 > /dev/null 2>&1
  done
done
python3 $R/tools/make_traffic.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE $TAG C2 > $O/${TAG}_pmc_traffic.json
python3 $R/tools/make_traffic.py --stats $O/${TAG}_kernel_stats_bench.csv $TAG C2 > $O/${TAG}_rocprofv3_averages.json
rm -rf $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
cp $R/profiles/traffic.json $O/${TAG}_traffic_full.json
cd $R
run_bench
ls $O | grep $TAG
# tsdr_group_*: four members on this one device, member threads forced -- do the members' uploads overlap? (memory-copy trace)
cd /tmp
for th in 0 2; do
  rm -rf $O/${TAG}_copytrace
  rocprofv3 --memory-copy-trace --output-format csv -d $O/${TAG}_copytrace -o run -- python3 $R/tools/group_devices.py 0,0,0,0 C2 threads=$th > $O/${TAG}_group4_threads$th.json 2> /dev/null
  f=$(find $O/${TAG}_copytrace -name '*memory_copy_trace.csv' | head -1)
  cp $f $O/${TAG}_group4_threads${th}_memory_copy_trace.csv
  echo "member_threads=$th: $(python3 $R/tools/copy_overlap.py $f)" >> $O/${TAG}_group4_copy_overlap.txt
  rm -rf $O/${TAG}_copytrace
done
cat $O/${TAG}_group4_copy_overlap.txt
