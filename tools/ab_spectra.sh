R=${GRAFT_REPO_ROOT:-.}
for round in 1 2 3; do
  for lib in ab/ntst.so cur; do
    if [ "$lib" = cur ]; then unset TSDR_HIP_LIB; else export TSDR_HIP_LIB=$R/$lib; fi
    echo "$lib $(python3 $R/bench.py --spectra-only waterfall --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-200)"
  done
done
