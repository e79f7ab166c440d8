"""Development aid: read a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and report, per kernel name, count / mean duration,
and how much of the traced span had 0 / 1 / >= 2 kernels in flight (do launches of different streams overlap?).
   python tools/trace_overlap.py <kernel_trace.csv> [name-substring to restrict the span to]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else None
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows]
ks.sort()
bursts = None
if sub:
    # the pipelined regions: bursts of kernels on queues other than the first library kernel's (the context's own stream
    # carries the non-pipelined legs of the same run); a gap of more than 0.5 ms ends a burst
    lib = [k for k in ks if "tsdr::" in k[2]]
    first_q = lib[0][3]
    lanes = [k for k in lib if k[3] != first_q]
    bursts = []
    for k in lanes:
        if bursts and k[0] - bursts[-1][1] < 500_000:
            bursts[-1][1] = max(bursts[-1][1], k[1])
        else:
            bursts.append([k[0], k[1]])
    ks = [k for k in lib if any(lo <= k[0] and k[1] <= hi for lo, hi in bursts)]
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
hist = collections.Counter()
if bursts is None:
    bursts = [[ev[0][0], ev[-1][0]]]
for lo, hi in bursts:
    depth, last = 0, lo
    for t, d in ev:
        if t < lo or t > hi:
            continue
        hist[min(depth, 2)] += t - last
        last = t; depth += d
tot = sum(hist.values())
print("pipelined span %.3f ms in %d burst(s): idle %.1f %%, one kernel %.1f %%, two or more %.1f %%" % (tot / 1e6, len(bursts), 100 * hist[0] / tot, 100 * hist[1] / tot, 100 * hist[2] / tot))
