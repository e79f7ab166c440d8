#!/bin/bash
# Evidence run for profiles/ (GPU box): the bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
# Usage (via gpurun): ./tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-r02_x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python3 $R/bench.py --quick --pipeline on > $O/${TAG}_bench_c2_pipeline.json 2>/dev/null
# round 4: kernel traces of the pipelined loop (do the image launch and the previous buffer's tail overlap?), the raster
# kernel's store-alignment A/B (option raster_split: 0 = one-launch walk, 1 = sheared raster-only + raster-free, 2 = unsheared)
$R/tools/prof_pipeline.sh $TAG > /dev/null 2>&1
$R/tools/ab_env2.sh TSDR_RASTER_SPLIT "0 1 2" > $O/${TAG}_raster_split_ab.txt 2>/dev/null
for v in 1 2; do
  rm -rf $O/${TAG}_split$v
  TSDR_RASTER_SPLIT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_split$v -o run -- python3 $R/bench.py --quick --no-pipeline-leg --steps 20 --repeats 2 > /dev/null 2>&1
  cp $(find $O/${TAG}_split$v -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_raster_split$v.csv
  rm -rf $O/${TAG}_split$v
done
$R/tools/calib_fetch.sh $TAG > /dev/null 2>&1
rm -rf $O/${TAG}_stats $O/${TAG}_pmc_*
# (the legs that run launches of two streams side by side -- `pipeline`, `two_streams` -- are left out of the profiled command:
# their stretched launches of the same kernel symbols would be averaged into the per-kernel durations the roofline is checked against)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o run -- python3 $R/bench.py --no-cpu --no-ingest --no-pipeline-leg --no-two-streams > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -o run -- python3 $R/bench.py --quick --steps 5 --warmup 2 --repeats 1 > /dev/null 2>&1
done
python3 $R/tools/make_traffic.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE $TAG C2 > $O/${TAG}_traffic.json
# GetSpectrum / resampler legs: one short profiled run per leg and counter (cold-cache cycling as in the bench)
CALLS=8
# search alone (kernel stats of the configuration search at C2)
rm -rf $O/${TAG}_search
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_search -o run -- python3 $R/tools/prof_search.py > /dev/null 2>&1
cp $(find $O/${TAG}_search -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_search_c2.csv 2>/dev/null
rm -rf $O/${TAG}_search
for leg in welch waterfall welch_1000 spectrum resampler_1024x4 resampler_1000000x4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${TAG}_pmcs_${leg}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmcs_${leg}_$c -o run -- python3 $R/bench.py --spectra-only $leg --steps $CALLS --warmup 0 > /dev/null 2>&1
  done
  python3 $R/tools/make_traffic.py --spectra $leg $CALLS $O/${TAG}_pmcs_${leg}_FETCH_SIZE $O/${TAG}_pmcs_${leg}_WRITE_SIZE $TAG >> $O/${TAG}_traffic_spectra.txt
  rm -rf $O/${TAG}_pmcs_${leg}_FETCH_SIZE $O/${TAG}_pmcs_${leg}_WRITE_SIZE
done
cp $R/profiles/traffic.json $O/${TAG}_traffic_full.json   # (the box's copy: C2 frame kernels + search + spectra legs of this tag)
# the bench line once more, now that this tag's traffic file is in place (its `traffic` fields are read from it)
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
ls $O | grep $TAG
