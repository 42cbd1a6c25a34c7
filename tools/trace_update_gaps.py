#!/usr/bin/env python3
"""Idle-gap accounting of ONE update in a rocprofv3 kernel trace (update = from one nchw_to_nhwc64 launch to the next).
usage: trace_update_gaps.py <kernel_trace.csv> <update index>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
marks = [i for i, e in enumerate(ev) if "nchw_to_nhwc64" in e[2]]
w = int(sys.argv[2])
lo, hi = ev[marks[w]][0], ev[marks[w + 1]][0]
tail = [e for e in ev if lo <= e[0] < hi]
end = max(e[1] for e in tail)
busy, cs, ce, gaps = 0, None, None, []
for s, e, n, q in tail:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
            gaps.append(s - ce)
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("launches %d  span %.3f ms  busy union %.3f ms  idle %.3f ms in %d gaps" % (len(tail), (end - lo) / 1e6, busy / 1e6, (end - lo - busy) / 1e6, len(gaps)))
h = collections.Counter()
for g in gaps:
    h[min(g // 1000, 20)] += g
print("idle by gap size (us bucket -> ms):", {int(k): round(v / 1e6, 3) for k, v in sorted(h.items())})
print("gap count by size:", dict(sorted(collections.Counter(min(g // 1000, 20) for g in gaps).items())))
