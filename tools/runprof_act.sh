# kernel statistics and one-replay timeline of the rollout-step graph at B=1: bash tools/runprof_act.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pa && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pa -- python3 tools/prof_act_graph.py > gpurun_out/pa.log 2>&1; echo rc=$?
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pa/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = 37
tot = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e3
print(f"kernel time per step {tot:.0f} us, launches per step {sum(int(r['Calls']) for r in rows) / n:.0f}")
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs']) / n / 1e3:8.1f} us {int(r['Calls']) / n:6.1f}/step  {r['Name'][:90]}")
PY
python3 tools/act_timeline.py $(ls gpurun_out/pa/*/*kernel_trace.csv | head -1) > gpurun_out/act_timeline.txt; tail -3 gpurun_out/act_timeline.txt
rm -rf gpurun_out/pa
