import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["sync_guard"], d.get("index_parity"), d["roofline"].get("kernels_ms_per_step") if "roofline" in d else None)
