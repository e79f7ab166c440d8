"""Development aid: read a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and report, per kernel name, count / mean duration,
and how much of the traced span had 0 / 1 / >= 2 kernels in flight (do launches of different streams overlap?).
   python tools/trace_overlap.py <kernel_trace.csv> [name-substring to restrict the span to]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else None
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows]
ks.sort()
if sub:
    # the pipelined region: kernels named `sub` that ran on a queue other than the one most of the run used
    import collections as _c
    main_q = _c.Counter(k[3] for k in ks if "tsdr::" in k[2] and sub not in k[2]).most_common(1)[0][0] if False else None
    lanes = _c.Counter(k[3] for k in ks if sub in k[2])
    laneq = [q for q, _ in lanes.most_common()][-1] if len(lanes) > 1 else list(lanes)[0]
    idx = [i for i, k in enumerate(ks) if sub in k[2] and k[3] == laneq]
    ks = ks[idx[len(idx) // 4]: idx[-1] + 1]   # steady state: skip the first quarter
    ks = [k for k in ks if "tsdr::" in k[2]]
agg = collections.OrderedDict()
for s, e, n, q in ks:
    n = n.split("(")[0][-60:]
    a = agg.setdefault(n, [0, 0, set()])
    a[0] += 1; a[1] += e - s; a[2].add(q)
for n, (c, t, q) in agg.items():
    print(f"{c:6d} x {t / c / 1e3:9.2f} us  queues {sorted(q)}  {n}")
ev = []
for s, e, n, q in ks:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[min(depth, 2)] += t - last
    last = t; depth += d
tot = sum(hist.values())
print("span %.3f ms: idle %.1f %%, one kernel %.1f %%, two or more %.1f %%" % (tot / 1e6, 100 * hist[0] / tot, 100 * hist[1] / tot, 100 * hist[2] / tot))
