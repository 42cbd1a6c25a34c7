timeout 600 python -m pytest tests -x -q -m gpu -k "conv or bf16 or golden or update or cfg4" 2>&1 | tail -3
for t in 0 1; do echo "== BM256=$t"; WSMG_CONV_BM256=$t timeout 200 python tools/bench_conv.py --dtype bf16 2>&1 | tail -18; done
