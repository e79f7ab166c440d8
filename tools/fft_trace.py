"""Per-kernel medians of the FFT launches in a rocprofv3 kernel trace CSV.   python tools/fft_trace.py <run_kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.OrderedDict()
for r in rows:
    if 'k_fft' not in r['Kernel_Name'] and 'k_ac' not in r['Kernel_Name']: continue
    key = (r['Kernel_Name'].split('(')[0].replace('void tsdr::', ''), r['Grid_Size_X'], r['Workgroup_Size_X'], r.get('VGPR_Count', ''), r.get('Scratch_Size', ''))
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for k, v in agg.items():
    v.sort(); print(f"{k[0]:28s} grid {k[1]:>9} wg {k[2]:>4} vgpr {k[3]:>4} scr {k[4]:>4}  n={len(v):3d} med {v[len(v)//2]:7.1f} min {v[0]:7.1f}")
