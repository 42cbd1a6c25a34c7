cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call16
mkdir -p $O
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-f32"
for i in 1 2; do
  for cfg in "WSMG_EARLY_RELAYOUT=1 WSMG_EARLY_PRIORITY=0" "WSMG_EARLY_RELAYOUT=1 WSMG_EARLY_PRIORITY=-1" "WSMG_EARLY_RELAYOUT=0 WSMG_EARLY_PRIORITY=-1" "WSMG_EARLY_RELAYOUT=0 WSMG_EARLY_PRIORITY=0"; do
    echo "== $cfg"; env $cfg WSMG_BENCH_WINDOW=20 timeout 300 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('sustained') or d['windows'])['ms_per_update_by_window'])"
  done
done | tee $O/early_relayout_ab2.txt
