#!/bin/bash
# Development aid (GPU box): bench.py --quick alternately on several library builds, the order rotated every round; per build the
# sorted per-launch times of the dominant kernel.   ROUNDS=5 BENCH_ARGS="..." tools/ab_rotate.sh cur ab/x.so ...
R=${GRAFT_REPO_ROOT:-.}
libs=("$@"); n=${#libs[@]}
for round in $(seq 0 $((${ROUNDS:-3}))); do
  for k in $(seq 0 $((n-1))); do
    lib=${libs[$(((k+round)%n))]}
    if [ "$lib" = cur ]; then unset TSDR_HIP_LIB; else export TSDR_HIP_LIB=$R/$lib; fi
    python3 $R/bench.py --quick $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$lib', d['ms_per_step'], round(r['avg_launch_ms']*1e3,1), r['kernel'])"
  done
done | python3 -c "
import sys,collections
t=collections.defaultdict(list); s=collections.defaultdict(list)
for l in sys.stdin:
    a=l.split(); t[a[0]+' '+a[3]].append(float(a[2])); s[a[0]+' '+a[3]].append(float(a[1]))
for k in t: print(k.ljust(40), 'kernel us sorted', sorted(t[k]), 'median step', sorted(s[k])[len(s[k])//2])"
