# interleaved A/B of environment settings on ONE box: bash tools/ab_pairs.sh "A_ENV=1" "B_ENV=2" [rounds] [steps]
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; R=${3:-3}; S=${4:-60}
run() { env $1 python3 bench.py --steps $S --warmup 5 --no-cpu-baseline --no-f32 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'], end=' ')"; }
for i in $(seq 1 $R); do echo -n "round $i:  [$A] "; run "$A"; echo -n "  [$B] "; run "$B"; echo; done
