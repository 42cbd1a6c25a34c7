#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection.csv files: per kernel (name substring filter), mean counter values per dispatch.
usage: pmc_fold.py <substr> <csv> [<csv> ...]"""
import csv, sys, collections
sub = sys.argv[1]
for path in sys.argv[2:]:
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        e = per.setdefault(d, {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if not per:
        continue
    keys = sorted(k for k in next(iter(per.values())) if k != "ns")
    n = len(per)
    print(path.split("/")[-3] if path.count("/") > 2 else path, "dispatches", n, "mean us %.1f" % (sum(e["ns"] for e in per.values()) / n / 1e3))
    for k in keys:
        print("   %-32s %16.0f" % (k, sum(e.get(k, 0.0) for e in per.values()) / n))
