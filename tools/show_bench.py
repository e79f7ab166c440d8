"""print the interesting parts of a bench.py JSON line (stdin or file)"""
import json, sys
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
d = json.loads([l for l in src if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "timing", "sync_margin", "search", "strong", "spectra"):
    print(k, d.get(k))
r = d["roofline"]
print("roofline", {k: r.get(k) for k in ("kernel", "frac", "avg_launch_ms", "traffic", "step_frac", "kernels_ms_per_step")})
for k in ("fused", "exact", "c5", "c3", "cpu_baseline", "host_ingest"):
    print(k, d.get(k))
