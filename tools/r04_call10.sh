# stall hunt: many short windows, collector plain / frozen heap / off, interleaved; + the new collate test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call10
mkdir -p $O
for i in 1 2 3; do
  for G in plain freeze 0; do
    echo "== GC=$G"; WSMG_BENCH_GC=$G WSMG_BENCH_WINDOW=5 WSMG_BENCH_HOSTTIME=1 timeout 300 python3 bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-f32 2>$O/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=(d.get('sustained') or d['windows'])['ms_per_update_by_window']; m=sorted(w)[len(w)//2]
print(d['ms_per_step'], 'median window', m, 'slow windows (>3%):', [x for x in w if x>1.03*m])"; grep "host enqueue" $O/err.txt
  done
done | tee $O/gc_ab.txt
