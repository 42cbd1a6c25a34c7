# per-kernel time of one update with / without the 3x3 window kernel (rocprofv3 kernel trace of bench.py)
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export WSMG_CONV_WIN3=$v
  OUT=$GRAFT_REPO_ROOT/gpurun_out/abk_$v
  rm -rf $OUT
  rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-f32 > $OUT.log 2>&1
  echo "WIN3=$v"; python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6 / 8)
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/8:8.3f} ms/update")
PY
done
