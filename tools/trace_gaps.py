#!/usr/bin/env python3
"""Fold a rocprofv3 --kernel-trace CSV of `bench.py` into: wall per update, GPU-busy union, idle gaps, per-queue
busy time, and the largest idle gaps with the kernels on either side.  usage: trace_gaps.py <kernel_trace.csv> <updates>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n_upd = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
# the timed updates are the tail of the trace: find update boundaries by the first launch of the entry kernel
marks = [i for i, e in enumerate(ev) if "nchw_to_nhwc64" in e[2]]
marks = marks[-n_upd:]
lo = ev[marks[0]][0]
tail = [e for e in ev if e[0] >= lo]
hi = max(e[1] for e in tail)
print("updates %d  launches/update %.0f  span %.3f ms/update" % (n_upd, len(tail) / n_upd, (hi - lo) / 1e6 / n_upd))
busy, cur_s, cur_e = 0, None, None
gaps = []
prev = None
for s, e, n, q in tail:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if prev is None or e >= cur_e:
        prev = n
busy += cur_e - cur_s
print("busy union %.3f ms/update   idle %.3f ms/update in %d gaps (%.1f us avg)" % (busy / 1e6 / n_upd, (hi - lo - busy) / 1e6 / n_upd, len(gaps), (hi - lo - busy) / 1e3 / max(1, len(gaps))))
perq = collections.Counter()
for s, e, n, q in tail:
    perq[q] += e - s
for q, t in perq.most_common():
    print("  queue %s: %.3f ms/update" % (q, t / 1e6 / n_upd))
hist = collections.Counter()
for g, a, b in gaps:
    hist[min(int(g / 1000), 50)] += g
print("idle by gap size (us bucket: ms/update):", {k: round(v / 1e6 / n_upd, 3) for k, v in sorted(hist.items())})
print("largest gaps:")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print("  %7.1f us  after %-50s before %s" % (g / 1e3, a[:50], b[:60]))

import re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:100]
mainq = perq.most_common(1)[0][0]
agg = collections.defaultdict(lambda: [0, 0])
for s_, e_, n, q in tail:
    if q == mainq:
        a = agg[short(n)]
        a[0] += e_ - s_; a[1] += 1
print("main queue, per update (ms, launches):")
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print("  %7.3f %5.1f  %s" % (t / 1e6 / n_upd, c / n_upd, n))
