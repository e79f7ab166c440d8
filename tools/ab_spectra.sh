# Development aid (GPU box): alternate one spectrum leg of bench.py between a library build and the in-tree one.   LEG=welch A=ab/x.so bash tools/ab_spectra.sh
R=${GRAFT_REPO_ROOT:-.}
for round in 1 2 3; do
  for lib in $A cur; do
    if [ "$lib" = cur ]; then unset TSDR_HIP_LIB; else export TSDR_HIP_LIB=$R/$lib; fi
    echo "$lib $(python3 $R/bench.py --spectra-only $LEG --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-200)"
  done
done
