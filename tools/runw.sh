timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -3
for tps in 1 2; do
echo "== TPS=$tps"; WSMG_CONV_TPS=$tps timeout 200 python tools/bench_conv.py --dtype bf16 2>&1 | tail -18
done
