#!/usr/bin/env python3
"""Fold one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE over `-- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32`) into
MFMA-pipe utilisation per kernel family.

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)

SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the chip's 1024 SIMDs (= 32 x the number of v_mfma_f32_32x32x16_bf16
issued, MI355X_MICROARCH.md "cycle constants"); GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs.  A kernel that
issues one MFMA per SIMD every 32 cycles reads 1.0, which at 2.4 GHz is the 2.5 PFLOP/s dense bf16 peak; achieved /
peak = mfma_busy x (effective clock / 2.4 GHz) x (algorithmic / issued FLOPs).  eff_clock_ghz = GRBM_GUI_ACTIVE / 8 /
dispatch duration (reads high on dispatches shorter than ~0.3 ms, same guide).  Families are folded cycle-weighted.
usage: pmc_mfma.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(conv_igemm_bf16_kernel<(true|false))", name)
    if m:
        return m.group(1) + ", *>"
    m = re.match(r"([A-Za-z_0-9:]+)", name)
    return m.group(1) if m else name[:40]


per = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    d = int(r["Dispatch_Id"])
    e = per.setdefault(d, {"name": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])

fam = collections.OrderedDict()
for d, e in per.items():
    f = fam.setdefault(family(e["name"]), collections.defaultdict(float))
    f["launches"] += 1
    f["ns"] += e["ns"]
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY",
              "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"):
        f[k] += e.get(k, 0.0)

# updates in the pass = the optimizer's launches / 3 (bench.py runs probe updates before --warmup / --steps since round 6)
UPDATES = max(1, sum(1 for e in per.values() if "adam_multi_kernel" in e["name"]) // 3)
res = {}
for name, f in fam.items():
    if f["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0:
        continue
    xcd_cycles = f["GRBM_GUI_ACTIVE"] / 8.0
    wave = max(f["SQ_WAVE_CYCLES"], 1.0)
    res[name] = {
        "launches_per_update": f["launches"] / UPDATES,
        "ms_per_update": f["ns"] / 1e6 / UPDATES,
        "mfma_busy": f["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * xcd_cycles),
        "eff_clock_ghz": xcd_cycles / f["ns"],
        "wave_time_split": {"issuing": f["SQ_ACTIVE_INST_ANY"] / wave, "issue_stalled": f["SQ_WAIT_INST_ANY"] / wave,
                            "of_which_lds": f["SQ_WAIT_INST_LDS"] / wave, "parked_waitcnt_or_barrier": f["SQ_WAIT_ANY"] / wave},
    }
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["ms_per_update"]))
json.dump({"note": __doc__.split("usage")[0].strip(), "kernels": res}, open(sys.argv[2], "w"), indent=1)
for n, v in res.items():
    print("%-44s %5.1f launches %7.3f ms/update  mfma_busy %.3f  clock %.2f GHz  lds-stall %.2f" % (
        n[:44], v["launches_per_update"], v["ms_per_update"], v["mfma_busy"], v["eff_clock_ghz"], v["wave_time_split"]["of_which_lds"]))
