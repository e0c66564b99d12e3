"""Reads a rocprofv3 kernel trace (csv) and reports, per pair of queues, how much of their kernels' time overlaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
print({q: len(v) for q, v in by_q.items()})
qs = sorted(by_q, key=lambda q: -len(by_q[q]))[:3]
def busy(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
    for s, e, *_ in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
# window: the last third of the trace (both at once)
t_all = [x for q in qs for x in by_q[q]]
t0, t1 = min(x[0] for x in t_all), max(x[1] for x in t_all)
lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
for q in qs:
    iv = [x for x in by_q[q] if x[0] >= lo]
    print("queue", q, "kernels", len(iv), "busy ms", busy(iv) / 1e6 if iv else 0, "sum of durations ms", sum(e - s for s, e, *_ in iv) / 1e6)
a = [x for x in by_q[qs[0]] if x[0] >= lo]; b = [x for x in by_q[qs[1]] if x[0] >= lo]
ua, ub, uab = busy(a), busy(b), busy(a + b)
print(f"busy A {ua/1e6:.2f} ms, busy B {ub/1e6:.2f} ms, union {uab/1e6:.2f} ms -> overlapped {(ua + ub - uab)/1e6:.2f} ms ({(ua + ub - uab) / min(ua, ub) * 100:.0f} % of the shorter)")
