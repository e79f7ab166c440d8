#!/bin/bash
# Development aid (GPU box): FFT and spectrum legs with two library builds alternating on one box.
#   tools/ab_micro.sh ab/<a>.so ab/<b>.so
cd /tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
for lib in "$@"; do
  echo "== $lib"
  TSDR_HIP_LIB=$R/$lib python3 $R/tools/time_fft.py 2000000 5000000 20000000 4194304 16777216 2>/dev/null | awk '{print $2, $3, $4, $5}'
  TSDR_HIP_LIB=$R/$lib python3 $R/tools/time_spectrum.py 2>/dev/null | head -3 | awk '{print $1,$2,$3,$4,$5,$6}'
done; done
