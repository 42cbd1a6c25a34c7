cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call11
mkdir -p $O
for i in 1 2 3 4; do
  WSMG_BENCH_WINDOW=5 WSMG_BENCH_HOSTTIME=2 timeout 300 python3 bench.py --steps 500 --warmup 5 --no-cpu-baseline --no-f32 2>$O/err$i.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=(d.get('sustained') or d['windows'])['ms_per_update_by_window']; m=sorted(w)[len(w)//2]
print(d['ms_per_step'], 'median window', m, 'slow windows (>3%):', [(i,x) for i,x in enumerate(w) if x>1.03*m])"; grep -E "host per update|collections" $O/err$i.txt | cut -c1-1500
done | tee $O/stalls.txt
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -k "collat" -x -q 2>&1 | tail -3
