#!/usr/bin/env python3
"""Timeline of ONE update from a rocprofv3 kernel trace of bench.py: every launch between two optimizer steps in start order with
duration, queue and the idle gap before it (no kernel of any queue running).  usage: update_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    return n[:96]
adam = [i for i, e in enumerate(ev) if "adam_multi_kernel" in e[2]]
# three adam launches per step: the update = the launches after the last launch of step k-1 up to the last launch of step k
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 5]
one = ev[ends[-2] + 1:ends[-1] + 1]
t0, busy_end = one[0][0], one[0][0]
qend = {}      # per queue: end of its previous launch ("own": how long this queue sat idle before the launch — waiting for another
               # queue's event, or for the host)
for s, e, n, q in one:
    gap = max(0, s - busy_end)
    own = (s - qend[q]) / 1e3 if q in qend else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  own {own:7.1f}  q{q}  {short(n)}")
    busy_end = max(busy_end, e)
    qend[q] = e
print(f"{len(one)} launches, span {(busy_end - t0) / 1e3:.0f} us")
