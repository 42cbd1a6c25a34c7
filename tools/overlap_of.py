#!/usr/bin/env python3
"""Kernels on OTHER hardware queues that overlap the launches of one kernel in a rocprofv3 kernel trace.
usage: overlap_of.py <kernel_trace.csv> <kernel substring>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    return n[:70]
hits = [e for e in ev if sys.argv[2] in e[2]]
for s, e, n, q in hits[-3:]:
    print(f"{short(n)} q{q} {(e - s) / 1e3:.1f} us")
    for s2, e2, n2, q2 in ev:
        if q2 != q and s2 < e and e2 > s:
            print(f"    q{q2} overlap {(min(e, e2) - max(s, s2)) / 1e3:7.1f} us of its {(e2 - s2) / 1e3:7.1f}  {short(n2)}")
