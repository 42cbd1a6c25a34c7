cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call12
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_round4.py -k "collate_emits or early_instruction" tests/test_gpu_kernels.py -k "collate_emits or early_instruction or collat or feeder" -x -q > $O/pytest.log 2>&1; echo pytest rc=$?; tail -12 $O/pytest.log
for i in 1 2 3; do
  for E in 1 0; do
  echo "== WSMG_EARLY_DEDUP=$E"; WSMG_EARLY_DEDUP=$E WSMG_BENCH_WINDOW=5 WSMG_BENCH_HOSTTIME=2 timeout 300 python3 bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-f32 2>$O/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=(d.get('sustained') or d['windows'])['ms_per_update_by_window']; m=sorted(w)[len(w)//2]
print(d['ms_per_step'], 'median window', m, 'slow windows (>3%):', [(i,x) for i,x in enumerate(w) if x>1.03*m])"; grep -E "host per update" $O/err.txt | cut -c1-400
  done
done | tee $O/early_dedup_ab.txt
