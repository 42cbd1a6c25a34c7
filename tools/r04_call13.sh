cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call13
mkdir -p $O
for i in 1 2; do
  for G in freeze plain; do
  echo "== WSMG_BENCH_GC=$G"; WSMG_BENCH_GC=$G WSMG_BENCH_WINDOW=5 WSMG_BENCH_HOSTTIME=2 timeout 300 python3 bench.py --steps 500 --warmup 5 --no-cpu-baseline --no-f32 2>$O/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=(d.get('sustained') or d['windows'])['ms_per_update_by_window']; m=sorted(w)[len(w)//2]
print(d['ms_per_step'], 'median window', m, 'slow windows (>3%):', [(i,x) for i,x in enumerate(w) if x>1.03*m])"; grep -E "collections" $O/err.txt | cut -c1-700
  done
done | tee $O/gc_freeze_500.txt
python3 -m cProfile -o $O/update.prof bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-f32 --prewarm-s 0 > /dev/null 2>&1
python3 -c "
import pstats; p=pstats.Stats('$O/update.prof'); p.sort_stats('tottime').print_stats(45)" > $O/cprofile_tottime.txt 2>&1
python3 -c "
import pstats; p=pstats.Stats('$O/update.prof'); p.sort_stats('cumtime').print_stats(70)" > $O/cprofile_cumtime.txt 2>&1
rm -f $O/update.prof
