# MFMA-busy and effective clock of one conv layer's kernels: bash tools/pmc_conv.sh <layer> <WSMG_CONV_WIN3 value>
cd /tmp && export TMPDIR=/tmp
export WSMG_CONV_WIN3=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv_$1_$2
rm -rf $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --dtype bf16 --reps 20 --only $1 > $OUT.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = {}
for r in csv.DictReader(open(f)):
    e = per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
fam = collections.defaultdict(lambda: collections.defaultdict(float))
for e in per.values():
    if e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0: continue
    k = e["name"][:60]
    fam[k]["n"] += 1; fam[k]["ns"] += e["ns"]
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        fam[k][c] += e.get(c, 0.0)
for k, v in fam.items():
    x = v["GRBM_GUI_ACTIVE"] / 8
    print(f"{k:60s} n={int(v['n']):4d} us={v['ns']/v['n']/1e3:7.1f} mfma_busy={v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*x):.3f} clk={x/v['ns']:.2f}GHz wait_any={v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.2f} wait_inst={v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:.2f} active={v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES']:.2f}")
PY
