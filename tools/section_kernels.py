#!/usr/bin/env python3
"""List every kernel of one update between two anchor kernels (by substring and occurrence) of a rocprofv3 kernel trace.
usage: section_kernels.py <trace.csv> <update index> <from substr> <from occurrence> <to substr> <to occurrence>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
marks = [i for i, e in enumerate(ev) if "nchw_to_nhwc64" in e[2]]
w = int(sys.argv[2])
lo, hi = ev[marks[w]][0], ev[marks[w + 1]][0]
upd = [e for e in ev if lo <= e[0] < hi]
def find(sub, occ):
    k = [e for e in upd if sub in e[2]]
    return k[int(occ)]
a, b = find(sys.argv[3], sys.argv[4]), find(sys.argv[5], sys.argv[6])
sec = [e for e in upd if a[1] <= e[0] < b[0]]
print("section %.1f us, %d kernels, kernel time sum %.1f us" % ((b[0] - a[1]) / 1e3, len(sec), sum(e[1] - e[0] for e in sec) / 1e3))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    return n[:110]
for s, e, n, q in sec:
    print("%8.1f +%6.1f q%s %s" % ((s - a[1]) / 1e3, (e - s) / 1e3, q, short(n)))
