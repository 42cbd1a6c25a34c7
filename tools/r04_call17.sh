cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call17
mkdir -p $O
B="python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-f32"
for i in 1 2 3; do
  for cfg in "WSMG_CATED_WGRAD_SIDE=1" "WSMG_CATED_WGRAD_SIDE=0"; do
    echo "== $cfg"; env $cfg timeout 300 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['windows']['ms_per_update_by_window'], d['loss'])"
  done
done | tee $O/cated_wgrad_side_ab.txt
