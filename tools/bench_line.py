import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"): continue
    d = json.loads(line); r = d["roofline"]
    print(sys.argv[1] if len(sys.argv) > 1 else "", d["config"]["precision"], "frames/s", d["value"], "ms/step", d["ms_per_step"], "| dom", r["kernel"], r["avg_launch_ms"], "ms", r["achieved"], "GB/s")
    print("     ", r["kernels_ms_per_step"])
    if d.get("search"): print("      search", d["search"])
    if d.get("fused"): print("      fused", d["fused"])
    if d.get("host_ingest"): print("      ingest", d["host_ingest"])
