# interleaved A/B of bench.py under one environment switch on ONE box: bash tools/ab_env.sh VAR A_VALUE B_VALUE [rounds]
VAR=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do
  for v in $A $B; do
    ms=$(env $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$VAR=$v $ms"
  done
done
