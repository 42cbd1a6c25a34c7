#!/usr/bin/env python3
"""Ordered kernel list of ONE update of a rocprofv3 kernel trace of bench.py (anchor: nchw_to_nhwc64 = first kernel of the
map stack).  usage: dump_update.py <kernel_trace.csv> <update index> > list.txt"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
anchor = sys.argv[3] if len(sys.argv) > 3 else "nchw_to_nhwc64"
marks = [i for i, e in enumerate(ev) if anchor in e[2]]
w = int(sys.argv[2])
lo, hi = ev[marks[w]][0], ev[marks[w + 1]][0]
upd = [e for e in ev if lo <= e[0] < hi]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:100]
qs = {}
for s, e, n, q in upd:
    qs.setdefault(q, [0, 0.0]); qs[q][0] += 1; qs[q][1] += (e - s) / 1e3
print("update span %.1f us, %d kernels; per queue (count, busy us): %s" % ((hi - lo) / 1e3, len(upd), {k: (v[0], round(v[1], 1)) for k, v in qs.items()}))
for s, e, n, q in upd:
    print("%9.1f +%7.1f q%s %s" % ((s - lo) / 1e3, (e - s) / 1e3, q, short(n)))
