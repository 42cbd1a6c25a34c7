#!/usr/bin/env python3
"""Timeline of ONE replay of the rollout-step graph from a rocprofv3 kernel trace (tools/runprof_act.sh): every launch in
start order with its duration and the idle gap before it (no kernel of any queue running), then the totals.
usage: act_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"at::native::", "", n)
    return n[:84]
# the last calls are separated by >= 10 ms of sleep (tools/prof_act_graph.py): split there, take the one before the last
groups, cur = [], [ev[0]]
for prev, e in zip(ev, ev[1:]):
    if e[0] - prev[1] > 5_000_000:
        groups.append(cur)
        cur = []
    cur.append(e)
groups.append(cur)
one = groups[-2]
per = len(one)
t0, busy_end, idle, busy = one[0][0], one[0][0], 0, 0
print(f"{per} launches per replay")
for s, e, n, q in one:
    gap = max(0, s - busy_end)
    idle += gap
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  q{q}  {short(n)}")
    busy_end = max(busy_end, e)
span = busy_end - t0
print(f"replay span {span / 1e3:.0f} us, idle inside {idle / 1e3:.0f} us, kernel sum {sum(e - s for s, e, _, _ in one) / 1e3:.0f} us")
