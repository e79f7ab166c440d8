#!/bin/bash
# rotate the order of the libraries every round; print per-library sorted kernel times at the end
R=${GRAFT_REPO_ROOT:-.}
libs=("$@"); n=${#libs[@]}
for round in $(seq 0 $((${ROUNDS:-9}))); do
  for k in $(seq 0 $((n-1))); do
    lib=${libs[$(((k+round)%n))]}
    if [ "$lib" = cur ]; then unset TSDR_HIP_LIB; else export TSDR_HIP_LIB=$R/$lib; fi
    python3 $R/bench.py --quick $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$lib', d['ms_per_step'], round(r['avg_launch_ms']*1e3,1))"
  done
done | python3 -c "
import sys,collections
t=collections.defaultdict(list); s=collections.defaultdict(list)
for l in sys.stdin:
    a=l.split(); t[a[0]].append(float(a[2])); s[a[0]].append(float(a[1]))
for k in t: print(k.ljust(16), 'kernel us sorted', sorted(t[k]), 'median step', sorted(s[k])[len(s[k])//2])"
