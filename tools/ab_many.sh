# several switches against their defaults, interleaved, on ONE box: bash tools/ab_many.sh
run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for i in 1 2 3; do
  echo "round $i: default $(run X=1) | WGRAD_STREAM=1 $(run WSMG_WGRAD_STREAM=1) | CONV_PF=2 $(run WSMG_CONV_PF=2) | DECODER_STREAMS=0 $(run WSMG_DECODER_STREAMS=0) | LSTM_AFTER_STEM=0 $(run WSMG_LSTM_AFTER_STEM=0) | WIN3=0 $(run WSMG_CONV_WIN3=0)"
done
