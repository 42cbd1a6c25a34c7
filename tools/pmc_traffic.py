#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each `-- python3 bench.py --steps 1 --warmup 1
--no-cpu-baseline --no-f32`) into HBM bytes per launch and per update for every kernel family.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), hence bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE).
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, json, re, sys, collections

def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(conv_igemm_bf16_kernel<(true|false))", name)
    if m:
        return m.group(1) + ", *>"
    m = re.match(r"([A-Za-z_0-9:]+)", name)
    return m.group(1) if m else name[:40]

def load(path, counter):
    rows = list(csv.DictReader(open(path)))
    per = collections.OrderedDict()
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        d = int(r["Dispatch_Id"])
        per.setdefault(d, [r["Kernel_Name"], 0.0])
        per[d][1] += float(r["Counter_Value"])
    return per

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
# updates in the pass = the optimizer's launches / 3 (bench.py runs probe updates before --warmup / --steps since round 6: a fixed 2 was wrong)
UPDATES = max(1, sum(1 for d in fetch if "adam_multi_kernel" in fetch[d][0]) // 3)
out = collections.OrderedDict()
for per, key in ((fetch, "fetch_kib"), (write, "write_kib")):
    for d in sorted(per):
        name, v = per[d]
        f = family(name)
        e = out.setdefault(f, {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
        e[key] += v
        if key == "fetch_kib":
            e["launches"] += 1
res = {}
for f, e in out.items():
    b = 1024.0 * (2.0 * e["fetch_kib"] + e["write_kib"])
    if e["launches"]:
        res[f] = {"launches_per_update": e["launches"] / UPDATES, "hbm_bytes_per_update": b / UPDATES,
                  "hbm_bytes_per_launch": b / e["launches"]}
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_update"]))
tot = sum(v["hbm_bytes_per_update"] for v in res.values())
json.dump({"note": __doc__.split("usage")[0].strip(), "total_hbm_bytes_per_update": tot, "kernels": res}, open(sys.argv[3], "w"), indent=1)
print("total GB/update %.2f" % (tot / 1e9))
for f, v in list(res.items())[:12]:
    print("%-50s %6.1f launches %8.3f GB/update %8.1f MB/launch" % (f[:50], v["launches_per_update"], v["hbm_bytes_per_update"] / 1e9, v["hbm_bytes_per_launch"] / 1e6))
