timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -3
for tile in "" 21 22 42; do
echo "== wgrad tile=$tile pf=1"; WSMG_WGRAD_TILE=$tile timeout 200 python tools/bench_conv.py --dtype bf16 2>&1 | awk '{print $1, $(NF-1), $NF}' | tail -17 | tr '\n' ';'; echo
done
