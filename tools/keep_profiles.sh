#!/bin/bash
# Copy the judged summaries of one evidence run (tools/collect_profiles.sh <tag>, tools/r05_full.sh <tag>) from the scratch
# directory gpurun_out/ into profiles/ (tracked).   tools/keep_profiles.sh <tag>
set -e
TAG=$1
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out; P=$R/profiles
cp $O/${TAG}_bench.json $P/${TAG}_bench.json
cp $O/${TAG}_bench_c2_pipeline.json $P/${TAG}_bench_c2_pipeline.json
cp $O/${TAG}_bench_under_rocprof.json $P/${TAG}_bench_under_rocprof.json
cp $O/${TAG}_stats/run_kernel_stats.csv $P/${TAG}_kernel_stats_bench.csv
cp $O/${TAG}_kernel_stats_search_c2.csv $P/${TAG}_kernel_stats_search_c2.csv
cp $O/${TAG}_traffic_full.json $P/${TAG}_pmc_traffic.json
cp $O/${TAG}_traffic_full.json $P/traffic.json
cp $O/${TAG}_fetch_calib.txt $P/${TAG}_fetch_calib.txt
for m in raster fused; do
  cp $O/${TAG}_pipe_${m}_overlap.txt $P/${TAG}_pipeline_${m}_overlap.txt
  cp $O/${TAG}_pipe_${m}_trace_excerpt.csv $P/${TAG}_pipeline_${m}_kernel_trace.csv
done
for f in sq_search sq_raster queue_probe time_fft_rows time_welch_sizes time_waterfall_sizes; do [ -f $O/${TAG}_$f.txt ] && cp $O/${TAG}_$f.txt $P/${TAG}_$f.txt; done
[ -f $O/${TAG}_fuzz_full.log ] && cp $O/${TAG}_fuzz_full.log $P/${TAG}_fuzz_full.log
[ -f $O/${TAG}_gputest.log ] && tail -12 $O/${TAG}_gputest.log > $P/${TAG}_gputest_tail.log
ls $P | grep $TAG
