"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel per dispatch."""
import csv, glob, sys, collections
for path in sorted(glob.glob(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_*/*/*_counter_collection.csv')):
    rows = list(csv.DictReader(open(path)))
    if not rows: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('tsdr::', '')[:28]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', path.split('/')[1])
    for k, d in agg.items():
        if not any(s in k for s in ('k_raster', 'k_down', 'k_sums', 'k_beta', 'k_shift', 'k_fir', 'k_fft', 'k_ac')): continue
        print(f"  {k:28s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
