#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of the calibration kernels of tools/ubench/fetch_calib.hip against their known byte counts.
#   tools/calib_fetch.sh <tag>      -> gpurun_out/<tag>_fetch_calib.txt
TAG=${1:-r04_x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/tools/ubench/fetch_calib.hip || exit 1
MIB=1024
: > $O/${TAG}_fetch_calib.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/calib_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/calib_$c -o run -- /tmp/fetch_calib $MIB > /tmp/calib_$c.out 2>/dev/null
  python3 - $c /tmp/calib_$c /tmp/calib_$c.out >> $O/${TAG}_fetch_calib.txt <<'PY'
import csv, glob, os, sys
c, d, outp = sys.argv[1:4]
known = {}
for line in open(outp):
    p = line.split()
    if len(p) >= 3 and p[1] == "bytes":
        known[p[0]] = int(p[2])
names = {"k_read<float>": "read4", "k_read<HIP_vector_type<float, 2": "read8", "k_read<HIP_vector_type<float, 4": "read16", "k_gather8": "gather8",
         "k_stage8": "stage8", "k_write<float>": "write4", "k_write<HIP_vector_type<float, 4": "write16"}
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != c:
            continue
        for sub, n in names.items():
            if sub in r["Kernel_Name"]:
                kb = float(r["Counter_Value"])
                print(f"{c:10s} {n:8s} counter {kb * 1024:14.0f} B   known {known.get(n, 0):14d} B   counter/known {kb * 1024 / max(known.get(n, 1), 1):.3f}")
PY
done
cat $O/${TAG}_fetch_calib.txt
