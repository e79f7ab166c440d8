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
# kernel traces of the pipelined loop in the arrangement the library measured fastest (do the launches overlap?)
$R/tools/prof_pipeline.sh $TAG > /dev/null 2>&1
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
# round 5: (a) SQ counters of the configuration search's FFT passes (what bounds each pass: VALU / LDS activity, waits);
# (b) the spectrum-path timing tools' outputs; (c) the pipeline against one call per buffer with K idle streams created first,
# hard-wired arrangement (TSDR_PIPE_TUNE=0: rounds 3-4) and measured one -- the hardware-queue dependence and its cure
$R/tools/pmc_search.sh k_fft > $O/${TAG}_sq_search.txt 2>&1
$R/tools/pmc_kernel.sh k_raster_fast cur > $O/${TAG}_sq_raster.txt 2>&1
for t in time_fft_rows time_welch_sizes time_waterfall_sizes; do
  timeout 600 python3 $R/tools/$t.py 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_$t.txt
done
{
for k in 0 1 5 6; do for tune in 0 1; do
  echo "== $k idle streams created first, TSDR_PIPE_TUNE=$tune"
  DUMMY_STREAMS=$k TSDR_PIPE_TUNE=$tune timeout 300 python3 $R/tools/time_pipeline.py 200 C2 2>&1 | grep "pipeline=\|measured"
done; done
} > $O/${TAG}_queue_probe.txt 2>&1
cp $R/profiles/traffic.json $O/${TAG}_traffic_full.json   # (the box's copy: C2 frame kernels + search + spectra legs of this tag)
# the bench line once more, now that this tag's traffic file is in place (its `traffic` fields are read from it)
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
ls $O | grep $TAG
