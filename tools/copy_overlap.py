"""Summarise a rocprofv3 --memory-copy-trace CSV: how much of the time the large host-to-device copies of a run were in flight
side by side (tsdr_group_*: one host thread per member).   python tools/copy_overlap.py <memory_copy_trace.csv>"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
big = []
for r in rows:
    d = (r.get("Direction") or r.get("Kind") or "").upper()
    if "HOST_TO_DEVICE" not in d and "H2D" not in d:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 100_000:      # > 0.1 ms: the members' slices, not the small state uploads
        big.append((s, e))
big.sort()
if not big:
    print("no large host-to-device copies found; columns:", list(rows[0].keys()) if rows else None)
    sys.exit(0)
ev = sorted([(s, 1) for s, _ in big] + [(e, -1) for _, e in big])
cur, last, t = 0, ev[0][0], {}
for ts, dlt in ev:
    t[cur] = t.get(cur, 0) + (ts - last)
    cur += dlt
    last = ts
busy = sum(v for k, v in t.items() if k >= 1)
print(f"{len(big)} host-to-device copies longer than 0.1 ms; time with >= 1 in flight {busy / 1e6:.2f} ms, of which "
      + ", ".join(f"{k} side by side {v / 1e6:.2f} ms ({100 * v / busy:.0f} %)" for k, v in sorted(t.items()) if k >= 1))
