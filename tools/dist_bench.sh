# per-update time distribution under environment settings: bash tools/dist_bench.sh "ENV=.." "ENV2=.." ...
cd $GRAFT_REPO_ROOT
for e in "$@"; do echo -n "[$e] "; env $e WSMG_BENCH_WINDOW=1 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-f32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); w=sorted(d['windows']['ms_per_update_by_window']); n=len(w)
print('mean %.3f  min %.2f p25 %.2f med %.2f p75 %.2f p90 %.2f max %.2f' % (d['ms_per_step'], w[0], w[n//4], w[n//2], w[3*n//4], w[9*n//10], w[-1]))"; done
