cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call15
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_round4.py -k "early or collate_emits or pipelined" -x -q > $O/pytest.log 2>&1; echo pytest rc=$?; tail -4 $O/pytest.log
B="python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-f32"
for i in 1 2 3; do
  for cfg in "WSMG_EARLY_RELAYOUT=1" "WSMG_EARLY_RELAYOUT=0"; do
    echo "== $cfg"; env $cfg timeout 300 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['windows']['ms_per_update_by_window'])"
  done
done | tee $O/early_relayout_ab.txt
