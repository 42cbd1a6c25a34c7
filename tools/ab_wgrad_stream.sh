run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2 3; do
  echo "round $i: off $(run WSMG_WGRAD_STREAM=0) | all $(run WSMG_WGRAD_STREAM=1) | small $(run WSMG_WGRAD_STREAM=2)"
done
