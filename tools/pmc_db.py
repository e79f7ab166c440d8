"""Summarise rocprofv3 rocpd (.db) counter collections: mean counter value per kernel per dispatch."""
import sqlite3, sys, glob, collections
pat = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_sq*/p_results.db'
flt = sys.argv[2:] or ['k_raster', 'k_down', 'k_proj', 'k_beta', 'k_shift', 'k_fft', 'k_ac']
for path in sorted(glob.glob(pat)):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kn = 'kernel_name' if 'kernel_name' in cols else 'name'
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, ctr, val in c.execute(f"select {kn}, counter_name, value from counters_collection"):
        k = name.split('(')[0].replace('void ', '').replace('tsdr::', '')[:40]
        agg[k][ctr].append(float(val))
    print('==', path)
    for k, d in agg.items():
        if not any(s in k for s in flt): continue
        print(f"  {k:40s} n={len(next(iter(d.values())))} " + "  ".join(f"{ct}={sum(v)/len(v):.4g}" for ct, v in sorted(d.items())))
