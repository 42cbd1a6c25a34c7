#!/usr/bin/env python3
"""Print the timeline (start, duration, queue) of the RNN / attention kernels of the last update in a rocprofv3
kernel trace.  usage: rnn_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
marks = [i for i, e in enumerate(ev) if "nchw_to_nhwc64" in e[2]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
lo = ev[marks[which]][0]
hi = ev[marks[which + 1]][0] if which + 1 < len(marks) and which != -1 else ev[-1][1]
t0 = None
for s, e, n, q in ev:
    if s < lo or s >= hi:
        continue
    if any(k in n for k in ("gru_", "lstm_", "attn_", "conv_igemm_bf16", "CatArray")):
        if t0 is None:
            t0 = s
        short = n.split("(")[0].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:34]
        print("%9.1f us  +%7.1f us  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short))
