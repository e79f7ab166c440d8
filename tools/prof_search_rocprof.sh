#!/bin/bash
# kernel trace of the configuration search (no per-kernel events): the launches' own durations and the gaps between them
cd /tmp && export TMPDIR=/tmp
N=${1:-4000000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/search_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $GRAFT_REPO_ROOT/tools/prof_search.py $N search > $OUT/run.log 2>&1
tail -12 $OUT/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/s_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "fft" in r["Kernel_Name"] or "amax" in r["Kernel_Name"]]
# the last 4 searches
tail = rows[-16:]
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r["Kernel_Name"][:60]:60s} dur={ (e-s)/1e3:7.2f} us gap={(s-prev)/1e3 if prev else 0:7.2f} us grid={r["Grid_Size_X"]} wg={r["Workgroup_Size_X"]} lds={r["LDS_Block_Size"]} vgpr={r.get("VGPR_Count")}')
    prev = e
PY
