#!/bin/bash
# GPU box: rocprofv3 kernel trace of the pipelined frame loop (raster-free, then raster) -> gpurun_out/<tag>_pipe_trace_*
TAG=${1:-r05_x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for mode in fused raster; do
  extra=""; [ $mode = fused ] && extra="--no-raster"
  rm -rf $O/${TAG}_pipe_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_pipe_$mode -o run -- python3 $R/bench.py --quick --pipeline on $extra --steps 300 --warmup 5 --repeats 2 > $O/${TAG}_pipe_${mode}_bench.json 2>/dev/null
  f=$(find $O/${TAG}_pipe_$mode -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/trace_overlap.py $f shift_iir > $O/${TAG}_pipe_${mode}_overlap.txt
  cat $O/${TAG}_pipe_${mode}_overlap.txt
  # keep the library's kernels only, compact (queue, start, end in ns from the first row, name)
  python3 - "$f" "$O/${TAG}_pipe_${mode}_trace.csv" <<'PY'
import sys, csv
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tsdr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["queue", "start_ns", "end_ns", "kernel"])
for r in rows:
    w.writerow([r["Queue_Id"], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"].split("(")[0].replace("void ", "")])
PY
  rm -rf $O/${TAG}_pipe_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_pipe_$mode -o run -- python3 $R/bench.py --quick --pipeline on $extra --steps 300 --warmup 5 --repeats 2 > $O/${TAG}_pipe_${mode}_bench.json 2>/dev/null
  f=$(find $O/${TAG}_pipe_$mode -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/trace_overlap.py $f shift_iir > $O/${TAG}_pipe_${mode}_overlap.txt
  cat $O/${TAG}_pipe_${mode}_overlap.txt
  # keep a 400-row excerpt of the steady state (the full trace is large)
  python3 - "$f" "$O/${TAG}_pipe_${mode}_trace_excerpt.csv" <<'PY'
import sys, csv
rows = list(csv.reader(open(sys.argv[1])))
h, b = rows[0], rows[1:]
mid = len(b) // 2
csv.writer(open(sys.argv[2], "w")).writerows([h] + b[mid:mid + 400])
PY
  rm -rf $O/${TAG}_pipe_$mode
done
