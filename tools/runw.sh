timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -3
for pf in 1 2 3; do
echo "== igemm/wgrad default tiles pf=$pf"; WSMG_CONV_PF=$pf timeout 200 python tools/bench_conv.py --dtype bf16 2>&1 | tail -18
done
for tile in 21 22 42; do for pf in 1 2; do
echo "== wgrad tile=$tile pf=$pf"; WSMG_WGRAD_TILE=$tile WSMG_CONV_PF=$pf timeout 200 python tools/bench_conv.py --dtype bf16 2>&1 | awk '{print $1, $(NF-1), $NF}' | tail -17 | tr '\n' ';'; echo
done; done
